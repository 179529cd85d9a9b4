"""ParmParse-compatible inputs (row f4): parsing rules and the key mapping, no GPU needed."""
import os

import pytest

from iamr_amd.inputs import Inputs, parse_text

HERE = os.path.dirname(os.path.abspath(__file__))
LDC = os.path.join(HERE, "golden", "inputs.3d.lid_driven_cavity16")


def test_parse_rules():
    t = parse_text('a.b = 1 2 3  # comment\n\n# only comment\nname = "two words" x\nc=4 d = 5\n')
    assert t == {"a.b": ["1", "2", "3"], "name": ["two words", "x"], "c": ["4"], "d": ["5"]}
    t = parse_text("a.b = 9\n", t)            # later definitions override
    assert t["a.b"] == ["9"]


def test_lid_driven_cavity_mapping_and_overrides():
    inp = Inputs([LDC], ["ns.cfl=0.25", "max_step = 2"])
    pr = inp.problem()
    assert pr["n"] == [16, 16, 16] and pr["periodic"] == [0, 0, 0] and pr["max_grid_size"] == 8
    p = pr["params"]
    assert p["cfl"] == 0.25 and p["init_dt"] == 0.0140625 and p["init_shrink"] == 0.3 and p["init_iter"] == 3
    assert p["visc_coef"] == 0.01 and p["tracer_diff_coef"] == 0.001
    assert p["phys_lo"] == [4, 4, 5] and p["phys_hi"] == [5, 5, 5]
    assert p["wall_vel_hi"][6:9] == [1.0, 0.0, 0.0] and sum(p["wall_vel_lo"]) == 0.0
    assert pr["prob"]["probtype"] == 1 and pr["max_step"] == 2
    assert set(inp.ignored) == {"amr.v"} and pr["plot_int"] > 0          # amr.plot_int drives the plotfile writer
    assert pr["check_int"] > 0 and pr["check_file"] == "chk" and pr["restart"] == ""      # amr.check_int drives the checkpoint writer


def test_unsupported_features_are_rejected_loudly():
    with pytest.raises(NotImplementedError):
        Inputs([LDC], ["amr.max_level=1"]).problem()
    with pytest.raises(NotImplementedError):
        Inputs([LDC], ["ns.lo_bc = 6 4 5"]).problem()          # not a physical BC type
    with pytest.raises(NotImplementedError):
        Inputs([LDC], ["prob.probtype=3"]).problem()
    with pytest.raises(KeyError):
        Inputs([LDC], ["ns.some_unknown_knob=1"]).problem()


def test_inflow_outflow_keys():
    pr = Inputs([LDC], ["ns.lo_bc = 1 4 5", "ns.hi_bc = 2 4 5", "xlo.velocity = 1. 0. 0.", "xlo.density = 1.", "xlo.tracer = 0.25"]).problem()
    p = pr["params"]
    assert p["phys_lo"] == [1, 4, 5] and p["phys_hi"] == [2, 4, 5]
    assert p["wall_vel_lo"][:3] == [1.0, 0.0, 0.0] and p["scal_bc_lo"][:2] == [1.0, 0.25] and p["scal_bc_hi"] == [1.0, 0.0, 0.0, 0.0] * 3   # [d*4 + slot]; defaults: density 1, tracer 0 (NavierStokes.cpp:70-84)


def test_rayleightaylor_keys():
    pr = Inputs([LDC], ["prob.probtype=10", "prob.rho_1=2.0", "prob.rho_2=1.0", "prob.tra_1=1.0", "prob.interface_width=0.05",
                        "prob.perturbation_amplitude=0.1", "ns.gravity=-9.8"]).problem()
    assert pr["prob"] == dict(probtype=10, rho_1=2.0, rho_2=1.0, tra_1=1.0, tra_2=0.0, pertamp=0.1, interface_width=0.05)
    assert pr["params"]["gravity"] == -9.8 and pr["params"]["do_mom_diff"] == 0 and pr["params"]["do_cons_trac"] == 0
    pr = Inputs([LDC], ["ns.do_mom_diff=1", "ns.do_cons_trac=1"]).problem()          # Exec/run3d/regtest.3d.rayleightaylor:6-7
    assert pr["params"]["do_mom_diff"] == 1 and pr["params"]["do_cons_trac"] == 1
    pr = Inputs([LDC], ["ns.do_temp=1", "ns.temp_cond_coef=1.e-8", "ns.do_trac2=1", "ns.do_cons_trac2=1", "ns.scal_diff_coefs=0.001 0.002",
                        "ns.lo_bc=1 5 5", "xlo.temp=2.0", "xlo.tracer2=0.5"]).problem()          # Exec/run3d/regtest.3d.hotspot:1-4
    p = pr["params"]
    assert (p["do_temp"], p["temp_cond_coef"], p["do_trac2"], p["do_cons_trac2"], p["tracer_diff_coef"], p["tracer2_diff_coef"]) == (1, 1.e-8, 1, 1, 0.001, 0.002)
    assert p["scal_bc_lo"][:4] == [1.0, 0.0, 0.5, 2.0]                      # density (default 1), tracer, tracer2, temp on x-lo: [d*4 + slot]
    with pytest.raises(NotImplementedError):
        Inputs([LDC], ["ns.do_LES=1"]).problem()


def test_host_side_problem_setups():
    """probtypes 4 / 5 / 7 (Source/prob/prob_init.cpp:232-281, :346-405, :562-610): keys, defaults (prob_init.H:9-35) and a few point values"""
    import numpy as np
    from iamr_amd.probinit import cell_centres, initial_state
    pr = Inputs([LDC], ["prob.probtype=7", "geometry.is_periodic=1 1 1", "ns.lo_bc=0 0 0", "ns.hi_bc=0 0 0"]).problem()
    assert pr["prob"]["probtype"] == 7 and pr["prob"]["density_ic"] == 1.0 and pr["prob"]["blob_radius"] == 0.1
    n = (32, 32, 32)
    X, Y, Z = cell_centres(n, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    assert X[0, 0, 0] == 0.5 / 32 and Z[0, 0, 31] == 31.5 / 32
    S = initial_state(pr["prob"], X, Y, Z)
    r = np.sqrt((Y - 0.5) ** 2 + (Z - 0.5) ** 2)
    assert np.allclose(S[..., 0], np.tanh((0.15 - r) / 0.0333)) and np.all(S[..., 1] == 0.0) and np.all(S[..., 3] == 1.0)
    assert abs(S[..., 2].max() - 0.05 * np.exp(-15.0 * 2 * (0.5 / 32) ** 2)) < 1e-15 and S[..., 4].max() <= 1.0
    assert np.array_equal(S[..., 0], S[::-1, ..., 0]) and np.array_equal(S[..., 0], S[:, ::-1, :, 0])       # tube along x, axisymmetric
    pr = Inputs([LDC], ["prob.probtype=5", "prob.direction=1", "prob.interface_width=1.0", "prob.blob_center=0.5 0.5 0.5", "prob.blob_radius=0.2"]).problem()
    S = initial_state(pr["prob"], X, Y, Z)
    assert np.allclose(S[..., 1], np.tanh(30.0 * (0.5 - X))) and np.allclose(S[..., 0], -0.05 * np.sin(np.pi * Y))
    assert set(np.unique(S[..., 4])) == {0.0, 1.0} and S[16, 16, 16, 4] == 1.0 and S[0, 0, 0, 4] == 0.0
    pr = Inputs([LDC], ["prob.probtype=4", "prob.velocity_ic=1. 0. 0.", "prob.blob_center=0.15 0.5 0.5", "prob.interface_width=0.001"]).problem()
    S = initial_state(pr["prob"], X, Y, Z)
    assert np.all(S[..., 0] == 1.0) and np.all(S[..., 1:3] == 0.0) and S[..., 4].min() == 0.0 and S[4, 16, 16, 4] == 1.0
    with pytest.raises(ValueError):
        initial_state(dict(pr["prob"], probtype=5, direction=2), X, Y, Z)


def test_fixed_grid_hierarchy_keys():
    """amr.max_level > 0 with amr.regrid_file (the fixed-grid runs of Exec/run2d/test_grids): boxes come out in each level's own index space;
    without a grid file or with a ratio other than 2 the run is refused"""
    amr_inp = os.path.join(HERE, "golden", "inputs.3d.taylorgreen_amr16")
    pr = Inputs([amr_inp]).problem()
    assert pr["fine_boxes"] == [[((4, 4, 4), (27, 27, 27))], [((20, 20, 20), (31, 43, 43)), ((32, 20, 20), (43, 43, 43))]]
    assert pr["params"]["visc_coef"] == 0.0 and pr["prob"]["probtype"] == 11
    with pytest.raises(NotImplementedError):
        Inputs([amr_inp], ["amr.ref_ratio = 4 2"]).problem()
    assert Inputs([amr_inp], ["ns.vel_visc_coef = 0.01"]).problem()["params"]["visc_coef"] == 0.01      # viscous hierarchies run
    with pytest.raises(NotImplementedError):
        Inputs([LDC], ["amr.max_level=1"]).problem()                       # no grid file: would need regridding


def test_physics_keys_are_not_swallowed_by_verbosity_prefixes():
    """`ns.v` must not match ns.variable_vel_visc / ns.visc_abs_tol / ns.vorterr ..."""
    # ns.do_reflux / ns.do_sync_proj / amr.subcycling_mode are read by the reference (NavierStokesBase.cpp:461-462);
    # the multi-level step here always refluxes, sync-projects and subcycles, so anything but their defaults is refused
    for k in ("ns.variable_vel_visc=1", "ns.variable_scal_diff=1", "ns.do_init_proj=0", "ns.do_mac_proj=0", "ns.do_reflux=0", "ns.do_sync_proj=0",
              "amr.subcycling_mode=None"):
        with pytest.raises(NotImplementedError):
            Inputs([LDC], [k]).problem()
    assert Inputs([LDC], ["amr.restart=chk00010"]).problem()["restart"] == "chk00010"      # checkpoint restart
    with pytest.raises(KeyError):
        Inputs([LDC], ["ns.vorterr=1.0"]).problem()
    assert Inputs([LDC], ["ns.variable_vel_visc=0", "ns.do_init_proj=1", "ns.v=1", "ns.do_reflux=1", "ns.do_sync_proj=1",
                          "amr.subcycling_mode=Auto"]).problem()["n"] == [16, 16, 16]


def test_refinement_indicator_keys():
    """amr.refinement_indicators (NS_error.cpp:10-108) -> the tagging rules of the regrid driver"""
    pr = Inputs([os.path.join(HERE, "golden", "inputs.3d.tracer_regrid16")]).problem()
    rg = pr["regrid"]
    assert pr["fine_boxes"] == [] and rg["max_level"] == 2 and rg["regrid_int"] == 2 and rg["blocking_factor"] == 4 and rg["n_error_buf"] == 1
    assert rg["rules"] == [dict(mode=0, value=[0.25], comp=4), dict(mode=0, value=[0.6, 0.6], comp=4)]
    with pytest.raises(KeyError):          # indicator sub-keys that no listed indicator uses are not silently dropped
        Inputs([os.path.join(HERE, "golden", "inputs.3d.tracer_regrid16")], ["amr.refinement_indicators = blob"]).problem()
    pr = Inputs([LDC], ["amr.max_level = 1", "amr.refinement_indicators = vort", "amr.vort.vorticity_greater = 5.0", "ns.vel_visc_coef = 0.01",
                 "amr.vort.max_level = 1", "amr.vort.in_box_lo = 0. 0. 0.", "amr.vort.in_box_hi = 1. 1. 0.5"]).problem()
    assert pr["regrid"]["rules"] == [dict(mode=2, value=[5.0], comp=-1, max_level=1, box_lo=[0.0, 0.0, 0.0], box_hi=[1.0, 1.0, 0.5])]


def test_two_dimensional_inputs_are_lifted_onto_a_slab():
    """judge row J2: the reference's 2-D inputs (Exec/run2d/regtest.2d.*, AMREX_SPACEDIM == 2 builds) run as y-periodic slabs of the 3-D
    library -- x -> x, y -> z (gravity and the hydrostatic outflow act along the last coordinate in both), cubic cells -- with the string
    BC types of NavierStokes::Initialize_bcs (NavierStokes.cpp:103-250)"""
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    pr = Inputs([os.path.join(here, "golden", "regtest.2d.poiseuille")]).problem()
    assert pr["slab"] == 8 and pr["n"] == [128, 8, 64] and pr["periodic"] == [0, 1, 0]
    assert pr["prob_lo"] == [0.0, 0.0, 0.0] and pr["prob_hi"] == [2.0, 8 * 2.0 / 128, 1.0]
    p = pr["params"]
    assert p["phys_lo"] == [1, 0, 5] and p["phys_hi"] == [2, 0, 5]          # xlo mass_inflow, xhi pressure_outflow, ylo / yhi nsw -> z faces
    assert p["wall_vel_lo"][:3] == [1.0, 0.0, 0.0] and p["gravity"] == 1.0 and p["use_forces_in_trans"] == 1 and p["use_ppm"] == 1
    assert p["scal_bc_lo"][:3] == [1.0, 0.0, 1.0] and p["do_trac2"] == 1 and p["do_cons_trac2"] == 1
    assert pr["prob"]["dim"] == 2 and pr["prob"]["blob_center"] == [0.15, 0.0, 0.5] and pr["prob"]["velocity_ic"] == [1.0, 0.0, 0.0]
    pr = Inputs([os.path.join(here, "golden", "regtest.2d.hotspot")]).problem()
    assert pr["n"] == [32, 8, 32] and pr["params"]["phys_lo"] == [5, 0, 4] and pr["params"]["phys_hi"] == [5, 0, 2] and pr["params"]["do_temp"] == 1
    # the 2-D initial data on the (x, z) plane: a hot bubble centred at (0, 0.65), uniform across the slab, no y-velocity
    import numpy as np
    from iamr_amd.probinit import cell_centres, initial_state
    X, Y, Z = cell_centres(pr["n"], pr["prob_lo"], pr["prob_hi"])
    S = initial_state(pr["prob"], X, Y, Z, 7)
    assert np.abs(S[:, 0] - S[:, 5]).max() == 0.0 and np.abs(S[..., 1]).max() == 0.0
    i, k = np.unravel_index(np.argmax(S[:, 0, :, 6]), S[:, 0, :, 6].shape)
    assert abs(X[i, 0, k]) < 0.07 and abs(Z[i, 0, k] - 0.65) < 0.07 and abs(S[i, 0, k, 6] * S[i, 0, k, 3] - 1.0) < 1e-14
    # string BC types: conflicts and nonsense are refused
    with pytest.raises(ValueError):
        Inputs([LDC], ["xlo.type=nsw"]).problem()                            # x is periodic in that file
    with pytest.raises(NotImplementedError):
        Inputs([os.path.join(here, "golden", "regtest.2d.poiseuille")], ["xhi.pressure=1.0"]).problem()


def test_plot_variable_selection():
    """amr.plot_vars / amr.derive_plot_vars (Amr::initPltAndChk): ALL / NONE / lists; defaults: every state variable, no derived one"""
    from iamr_amd.inputs import Inputs, parse_text
    from iamr_amd.plotfile import plot_selection, state_names
    st = state_names(1, 0)
    assert plot_selection(st) == (list(range(6)), [])
    assert plot_selection(st, "ALL", "ALL") == (list(range(6)), ["energy", "mag_vort", "avg_pressure"])
    assert plot_selection(st, ["tracer2", "density"], ["mag_vort"]) == ([3, 5], ["mag_vort"])
    assert plot_selection(st, "NONE", ["avg_pressure", "energy"]) == ([], ["avg_pressure", "energy"])
    with pytest.raises(ValueError):
        plot_selection(st, ["temp"])
    with pytest.raises(ValueError):
        plot_selection(st, "ALL", ["vorticity"])
    inp = Inputs.__new__(Inputs)
    inp.table, inp.used, inp.ignored = parse_text("amr.plot_vars = density tracer\namr.derive_plot_vars = ALL\n"), set(), []
    assert inp.name_list("amr.plot_vars", "ALL") == ["density", "tracer"] and inp.name_list("amr.derive_plot_vars", "NONE") == "ALL"
    assert inp.name_list("amr.absent", "NONE") == "NONE"


def test_c3_inputs_lift_onto_the_slab():
    """tests/golden/inputs.2d.doubleshearlayer_c3 (bench.py's config C3 line): a 2-D file -> x, slab (y, periodic, 8 cells), z"""
    import os
    from iamr_amd.inputs import Inputs
    inp = Inputs([os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inputs.2d.doubleshearlayer_c3")], ["amr.n_cell=512 512"])
    pr = inp.problem()
    assert pr["n"] == [512, 8, 512] and pr["slab"] == 8 and pr["periodic"] == [1, 1, 1]
    assert pr["regrid"]["max_level"] == 1 and pr["regrid"]["regrid_int"] == 1 and pr["prob"]["probtype"] == 5
    assert abs(pr["prob_hi"][1] - 8 * 2.0 / 512) < 1e-15
