"""the reference's solvers live in AMReX / AMReX-Hydro, which are absent; the oracle restates their published
algorithms and the product is free in the choices that cannot be pinned on reference output -- the multigrid cycle shape (sweeps per
smooth call, pre/post smooth calls), the GSRB over-relaxation factor, the bottom-solver tolerance and the smoother of the nodal solver.
None of them may change a converged answer beyond the solver tolerances: this test runs the same TaylorGreen steps under each
choice and bounds the spread of the states (and of the pressure, up to the additive constant of the singular periodic system)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

VARIANTS = {
    "upstream nodal cycle (4 sweeps, 2+2)": dict(nodal_sweeps=4, nodal_nu1=2, nodal_nu2=2),
    "host-driven BiCGStab on the 2^3 level (upstream hierarchy)": dict(device_bottom=0),
    "nodal 1 sweep, 2+2": dict(nodal_sweeps=1, nodal_nu1=2, nodal_nu2=2),
    "nodal Jacobi smoother": dict(nodal_smoother=2, nodal_sweeps=4, nodal_nu1=2, nodal_nu2=2),
    "GSRB omega 1.0": dict(omega=1.0),
    "GSRB omega 1.3": dict(omega=1.3),
    "cell-centred V(3,3)": dict(nu1=3, nu2=3),
    "bottom rtol 1e-8": dict(bottom_reltol=1e-8),
    "smoother as bottom solver": dict(bottom_smoother_only=1),
}
# process-wide switches read once per process cannot be flipped inside one test: the initial guesses of the MAC / nodal solves (previous or
# time-extrapolated potential instead of upstream's zero, IAMRX_WARM_START / IAMRX_WARM_EXTRAP) are covered by every parity test against the
# oracle, which always starts from zero.


def _run(lib, N, opts_kw, boxes):
    n = (16,) * 3
    g = lib.Geom.make(n)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.5, visc_coef=1e-2, init_iter=2), lib.mg_opts(**opts_kw))
    ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
    ns.post_init(-1.0)
    dts = [ns.step() for _ in range(3)]
    S = ns.data(N.NavierStokes.S_NEW).gather_valid(n)
    P = ns.data(N.NavierStokes.P_NEW).gather_valid(n)[..., 0]
    Gp = ns.data(N.NavierStokes.GP_NEW).gather_valid(n)
    return dts, S, P - P.mean(), Gp


@pytest.mark.parametrize("boxes", [None, 8])
def test_converged_answers_do_not_depend_on_unpinned_solver_choices(gpu, boxes):
    lib = gpu
    from iamr_amd import ns as N
    dts0, S0, P0, G0 = _run(lib, N, {}, boxes)
    for name, kw in VARIANTS.items():
        dts, S, P, G = _run(lib, N, kw, boxes)
        assert np.allclose(dts, dts0, rtol=1e-9, atol=0.0), name
        # solver tolerances: 1e-12 (projections), 1e-11 x norm (viscous); three steps + init iterations accumulate to < 1e-9
        assert np.abs(S - S0).max() <= 1e-9, (name, np.abs(S - S0).max())
        assert np.abs(G - G0).max() <= 1e-7 * max(1.0, np.abs(G0).max()), (name, np.abs(G - G0).max())
        assert np.abs(P - P0).max() <= 1e-7 * max(1.0, np.abs(P0).max()), (name, np.abs(P - P0).max())


KERNEL_FORMS = {
    # the finest-level smoother / residual read the stored face coefficients (rounds 1-2) instead of recomputing them from the
    # cell-centred density (MAC) or taking the three constants of a constant-viscosity operator
    "face coefficient arrays": {"ABEC_SIG": 0},
    # the general colour kernel instead of the lean (k_abec_gsrb1) and pair-marching (k_abec_gsrb2) ones
    "general colour kernel": {"GSRB2": 0, "GSRB1_NP": 0},
    "lean kernel, two planes in flight": {"GSRB2": 0, "GSRB1_NP": 2},
    "lean kernel, four planes in flight": {"GSRB2": 0, "GSRB1_NP": 4},
    "pair-marching, 4 planes per thread": {"GSRB2_TZ": 4},
    # residual and restriction as two passes, the residual of the convergence test by the one-cell-per-thread kernel, the correction
    # zeroed by a fill instead of by the first colour pass
    "unfused down-leg": {"RESID_RESTRICT": 0, "RESID_PAIRS": 0, "GSRB_ZERO": 0},
    "fused restriction only": {"RESID_PAIRS": 0},
    # tensor residual as two launches (7-point part, then the cross terms) instead of one
    "unfused tensor residual": {"TENSOR_FUSED": 0},
    # round 4: the three components of a constant-coefficient tensor colour pass as three pair-marching launches instead of one launch of
    # the general kernel
    "multi-component colour pass per component": {"GSRB2_MULTI": 1},
    # round 4: two colour passes instead of the one-launch red + black sweep (index-wrap levels with 128 / 256 cells in x)
    "two colour passes": {"GSRB_RB": 0},
    # round 4: ghost fills in front of the pair-marching residual kernels instead of index wrap
    "residual kernels read ghost cells": {"RESID_WRAP": 0},
    "two colour passes on levels with walls": {"GSRB_RB_WALLS": 0},
    # round 4: the coarser levels of a constant-coefficient operator read their (constant) coefficient arrays instead of taking the constants
    "coefficient arrays on the coarser levels": {"MG_COARSE_UNIFORM": 0},
    # round 5: the last two levels of a V-cycle in the one single-workgroup launch k_abec_tail instead of their ~14 separate launches
    # (opt-in: measured slower than the launches it replaces)
    "fused coarse tail": {"MG_TAIL_FUSED": 1},
    "fused coarse tail, coefficient arrays": {"MG_TAIL_FUSED": 1, "MG_COARSE_UNIFORM": 0},
    # round 5: `sol += cor` as its own pass behind the V-cycle instead of inside the last sweep of the sweep kernel (ACC)
    "sol += cor as its own pass": {"MG_ACC_LAST_SWEEP": 0},
}
DEFAULTS = {"ABEC_SIG": 1, "GSRB2": 1, "GSRB1_NP": 1, "GSRB2_TZ": 32, "RESID_RESTRICT": 1, "RESID_PAIRS": 1, "GSRB_ZERO": 1, "TENSOR_FUSED": 1, "GSRB2_MULTI": 0, "GSRB_RB": 1, "RESID_WRAP": 1, "GSRB_RB_WALLS": 1, "MG_COARSE_UNIFORM": 1, "MG_TAIL_FUSED": 0, "MG_ACC_LAST_SWEEP": 1}


@pytest.mark.parametrize("case", ["periodic_boxes", "periodic_one_box", "periodic_long_box", "channel_walls", "channel_walls_long"])
def test_kernel_forms_of_the_cell_centred_multigrid_give_the_same_doubles(gpu, case):
    """Round 3 replaced the kernels of the cell-centred multigrid's finest level by forms that read less (coefficients recomputed from the
    cell-centred density or taken as constants, pair-marching colour pass); each keeps the expressions of the kernel it replaces, so a run
    under any of them is the same run bit for bit -- states, pressure, time steps.  Removing the residual mean of a singular system in
    front of every cycle (upstream) instead of the first one only changes round-off."""
    lib = gpu
    from iamr_amd import ns as N

    def run():
        if case.startswith("periodic"):        # variable density, viscous, diffusive tracer; 8 boxes (ghost exchanges between the colour passes)
            n = (32, 16, 16)                    # or one box spanning the domain (wrap kernels, zero-fill pass, fused tensor residual: 32 x 16 tiles)
            if case == "periodic_long_box":     # 128 cells in x: the one-launch red + black sweep (k_abec_gsrb_rb) takes the finest level of the
                n = (128, 16, 16)               # MAC solve (density form) and of the three-component viscous solve (constants, a-term)
            g = lib.Geom.make(n, prob_hi=(n[0] / 16.0, 1.0, 1.0))
            lay = lib.Layout.decompose(n, 16) if case == "periodic_boxes" else lib.Layout.single(n)
            ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.5, visc_coef=2e-2, tracer_diff_coef=1e-2, init_iter=1))
            ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
            m = ns.data(N.NavierStokes.S_NEW)
            G = np.zeros(tuple(v + 2 for v in n) + (5,), order="F")
            G[1:-1, 1:-1, 1:-1] = m.gather_valid(n)
            x = (np.arange(n[0]) + 0.5) / n[0]
            G[1:-1, 1:-1, 1:-1, 3] = 1.0 + 0.3 * np.sin(2 * np.pi * x)[:, None, None]
            m.set_from_global(G, (-1, -1, -1))
            ns.set_data(N.NavierStokes.S_NEW, m)
        else:                                   # inflow / outflow in x, no-slip walls in y, slip in z: Dirichlet and Neumann faces, odd box
            n = (24, 12, 8)                     # lengths on the coarser multigrid levels
            if case == "channel_walls_long":    # 128 cells in x: the one-launch sweep with domain walls (Neumann / Dirichlet faces of the MAC
                n = (128, 16, 16)               # solve, no-slip / slip / outflow faces of the three viscous components)
            g = lib.Geom.make(n, prob_hi=(n[0] / 12.0 if n[0] > 24 else 2.0, 1.0, 0.5), periodic=(0, 0, 0))
            lay = lib.Layout.single(n)
            wl = [0.0] * 9
            wl[0] = 1.0
            sl = [0.0] * 12
            sl[0] = 1.0
            ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.5, visc_coef=0.05, init_iter=1, init_shrink=0.3, phys_lo=[1, 5, 4], phys_hi=[2, 5, 4],
                                                    wall_vel_lo=wl, scal_bc_lo=sl))
            ns.init_rest(1.0)
            m = lib.MultiFab(lay, lib.CELL, 5, 1)
            G = np.zeros(tuple(v + 2 for v in n) + (5,), order="F")
            G[1:-1, 1:-1, 1:-1, 0] = 1.0
            G[1:-1, 1:-1, 1:-1, 3] = 1.0
            m.set_from_global(G, (-1, -1, -1))
            ns.set_data(N.NavierStokes.S_NEW, m)
        ns.post_init(-1.0)
        dts = [ns.step() for _ in range(2)]
        return dts, ns.data(N.NavierStokes.S_NEW).gather_valid(n), ns.data(N.NavierStokes.P_NEW).gather_valid(n)[..., 0]

    # round 5: the constant-viscosity tensor residual runs in its cell-centred form (k_tensor_uni: mixed second differences, another
    # summation order than the five face fluxes) -- the bit-for-bit family below is that of the face-flux kernels; the cell-centred form is
    # compared with it to round-off afterwards
    # round 5 as well: the coarsest level of a constant-viscosity tensor solve is solved directly (k_dense_bottom) -- a bottom solver, one of
    # the unpinned choices, not a kernel form: the family below keeps the Krylov bottom solver (the forms that switch the constants off would
    # fall back to it anyway); the direct one is compared with it to the solver tolerance afterwards and in tests/test_gpu_tensor_bottom.py
    cc_old = lib.tuning_get("TENSOR_UNI_CC", 1)
    tb_old = lib.tuning_get("TENSOR_BOTTOM_DIRECT", 1)
    lib.tuning_set("TENSOR_UNI_CC", 0)
    lib.tuning_set("TENSOR_BOTTOM_DIRECT", 0)
    try:
        dts0, S0, P0 = run()
        for name, keys in KERNEL_FORMS.items():
            old = {k: lib.tuning_get(k, -1.0) for k in keys}
            try:
                for k, v in keys.items():
                    lib.tuning_set(k, v)
                dts, S, P = run()
            finally:
                for k, v in old.items():
                    lib.tuning_set(k, DEFAULTS[k] if v < 0 else v)
            assert dts == dts0, name
            assert np.array_equal(S, S0), (name, np.abs(S - S0).max())
            assert np.array_equal(P, P0), (name, np.abs(P - P0).max())
        lib.tuning_set("TENSOR_UNI_CC", 1)
        dts, S, P = run()
        assert np.allclose(dts, dts0, rtol=1e-12, atol=0.0)
        assert np.abs(S - S0).max() <= 1e-11 and np.abs(P - P0).max() <= 1e-9 * max(1.0, np.abs(P0).max()), (np.abs(S - S0).max(), np.abs(P - P0).max())
        lib.tuning_set("TENSOR_BOTTOM_DIRECT", 1)
        dts, S, P = run()
        assert np.allclose(dts, dts0, rtol=1e-9, atol=0.0)
        assert np.abs(S - S0).max() <= 1e-9 and np.abs(P - P0).max() <= 1e-7 * max(1.0, np.abs(P0).max()), (np.abs(S - S0).max(), np.abs(P - P0).max())
    finally:
        lib.tuning_set("TENSOR_UNI_CC", cc_old)
        lib.tuning_set("TENSOR_BOTTOM_DIRECT", tb_old)
    lib.tuning_set("MG_RES_MEAN", 1)
    try:
        dts, S, P = run()
    finally:
        lib.tuning_set("MG_RES_MEAN", 0)
    assert np.allclose(dts, dts0, rtol=1e-9, atol=0.0)
    assert np.abs(S - S0).max() <= 1e-9
