"""the multi-level time step (subcycled advance of a refined level, flux / sync registers, reflux, average down,
MAC sync, MLsyncProject, multi-level initialisation) -- HIP hierarchy (iamrx_amr_*, through the C-ABI) against the CPU oracle
(oracle/orc_amr.c), which solves the composite nodal systems with a different solver (conjugate gradients on the conforming
composite operator) and holds every refined level as a whole-domain array.

Tolerances: the composite sync solves stop at sync_tol = 1e-10 (relative), the level solves at 1e-12; the states of the two codes
agree to <= 2e-8 of the velocity scale after full coarse steps."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


def _make(n0, fine_boxes, crse_split, params_kw, state_fn, finer=(), periodic=(1, 1, 1)):
    from iamr_amd import lib as L
    from iamr_amd.ns import ns_params
    from iamr_amd.amr import Amr
    L.init()
    g0 = L.Geom.make([n0] * 3, periodic=periodic)
    lay0 = L.Layout.decompose([n0] * 3, crse_split)
    boxes = [fine_boxes] + list(finer)
    lays = [lay0] + [L.Layout([(tuple(lo), tuple(hi)) for lo, hi in b]) for b in boxes]
    amr = Amr(g0, lays, ns_params(**params_kw), L.mg_opts())
    og = orc.geom([n0] * 3, periodic=periodic)
    oa = orc.OrcAmr(og, orc.ns_params(**params_kw), orc.mg_opts(), [[]] + boxes)
    for l in range(len(lays)):
        S = state_fn(*oa.cell_centres(l))
        oa.set_state(l, S)
        lev = amr.levels[l]
        mf = L.MultiFab(lev.layout, L.CELL, S.shape[-1], 1)
        G = np.zeros(tuple(s + 2 for s in S.shape[:3]) + (S.shape[-1],), order="F")
        G[1:-1, 1:-1, 1:-1] = S
        mf.set_from_global(G, (-1, -1, -1))
        lev.set_data(lev.S_NEW, mf)
    return amr, oa


def _compare(amr, oa, tol, tag):
    worst = 0.0
    # every face periodic or a wall: the nodal systems are singular and the additive constant of the pressure is whatever the multigrid
    # iteration leaves (amrex::MLMG does not normalise it either); it depends on the cycle shape, which differs between the product
    # (iamrx_mg_opts defaults) and the oracle (upstream shape).  One constant for the whole hierarchy, taken from level 0.
    p_shift = None
    for l in range(oa.nlev):
        lev = amr.levels[l]
        n = oa.n(l)
        cov = oa.cov(l)
        S = lev.data(lev.S_NEW).gather_valid(n)
        So = oa.state(l)
        scale = max(1.0, np.abs(So[cov]).max())
        err = np.abs(S - So)[cov].max() / scale
        assert err <= tol, f"{tag}: level {l} state differs by {err}"
        worst = max(worst, err)
        Gp = lev.data(lev.GP_NEW).gather_valid(n)
        Go = oa.fab(l, 4).valid(n)
        gs = max(1.0, np.abs(Go[cov]).max())
        gerr = np.abs(Gp - Go)[cov].max() / gs
        assert gerr <= 50 * tol, f"{tag}: level {l} grad p differs by {gerr}"
        # pressure on the nodes of the level's cells
        P = lev.data(lev.P_NEW).gather_valid(n)[..., 0]
        Po = oa.fab(l, 2).valid(n, orc.NODE)[..., 0]
        nm = np.zeros(P.shape, bool)
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    nm[dx:dx + n[0], dy:dy + n[1], dz:dz + n[2]] |= cov
        d = (P - Po)[nm]
        if p_shift is None:
            p_shift = d.mean()
        d = d - p_shift
        perr = np.abs(d).max() / max(1.0, np.abs(Po[nm]).max())
        assert perr <= 100 * tol, f"{tag}: level {l} pressure differs by {perr}"
    return worst


def _composite_sum(amr_levels_state, cov1, dx0, dx1, comp):
    S0, S1 = amr_levels_state
    covc = cov1[::2, ::2, ::2]
    return (S0[..., comp] * ~covc).sum() * np.prod(dx0) + (S1[..., comp] * cov1).sum() * np.prod(dx1)


def _two_tracer_state(X, Y, Z):
    """Taylor-Green velocity, a density that varies, the usual tracer and a second one (ns.do_trac2: state component 5)"""
    S5 = orc.taylorgreen_state(X, Y, Z, c=1.0)
    S = np.zeros(S5.shape[:3] + (6,), order="F")
    S[..., :5] = S5
    tp = 2.0 * np.pi
    S[..., 3] = 1.0 + 0.2 * np.sin(tp * Y) * np.cos(tp * X)
    S[..., 5] = 1.0 + 0.5 * np.sin(tp * X) * np.cos(tp * Z) + 0.25 * np.cos(2.0 * tp * Y)
    return S


@pytest.mark.parametrize("case", ["one_box", "l_shape", "viscous", "viscous_l_shape_cons", "two_tracers", "two_tracers_cons"])
def test_two_level_taylorgreen_matches_oracle(case):
    n0 = 16
    kw = dict(cfl=0.7, visc_coef=0.0, init_iter=2)
    state_fn = lambda X, Y, Z: orc.taylorgreen_state(X, Y, Z, c=1.0)
    ncomp = 5
    if case.startswith("two_tracers"):
        # ns.do_trac2 (regtest.3d.poiseuille, regtest.3d.hotspot): NUM_STATE = 6, both tracers diffusive with their own coefficients, one
        # advected convectively (Laplacian_S) and one conservatively (Laplacian_SoverRho); variable density; viscous
        fine = [([8, 8, 8], [15, 15, 23]), ([16, 8, 8], [23, 15, 23]), ([8, 16, 8], [15, 23, 23])]
        split = 8
        cons = case.endswith("cons")
        kw.update(visc_coef=0.01, tracer_diff_coef=0.005, do_trac2=1, do_cons_trac2=0 if cons else 1, do_cons_trac=1 if cons else 0,
                  tracer2_diff_coef=0.01, do_mom_diff=1 if cons else 0)
        state_fn = _two_tracer_state
        ncomp = 6
    elif case in ("one_box", "viscous"):
        fine = [([4, 4, 4], [19, 19, 19])]
        split = 16
        if case == "viscous":
            # viscous / diffusive hierarchy: tensor solve with coarse/fine faces on the refined level, viscous flux registers, viscous terms in
            # the sync forcing, diffuse_Vsync / the scalar sync solve in mac_sync
            kw.update(visc_coef=0.01, tracer_diff_coef=0.005)
    elif case == "viscous_l_shape_cons":
        fine = [([8, 8, 8], [15, 15, 23]), ([16, 8, 8], [23, 15, 23]), ([8, 16, 8], [15, 23, 23])]
        split = 8
        kw.update(visc_coef=0.02, tracer_diff_coef=0.01, do_mom_diff=1, do_cons_trac=1, be_cn_theta=1.0)
    else:
        # L-shaped refined region made of three boxes (the shape of Exec/run2d/test_grids/fixed_grids_2), coarse level in 8 boxes
        fine = [([8, 8, 8], [15, 15, 23]), ([16, 8, 8], [23, 15, 23]), ([8, 16, 8], [15, 23, 23])]
        split = 8
    amr, oa = _make(n0, fine, split, kw, state_fn)
    amr.post_init()
    oa.post_init()
    assert abs(amr.dts()[0] - oa.dt(0)) <= 1e-9 * oa.dt(0) and abs(amr.dts()[1] - oa.dt(1)) <= 1e-9 * oa.dt(1)
    _compare(amr, oa, 2e-8, "after post_init")
    cov1 = oa.cov(1)
    m0 = [_composite_sum((oa.state(0), oa.state(1)), cov1, oa.dx(0), oa.dx(1), c) for c in range(ncomp)]
    for step in range(2):
        dt = amr.coarse_step()
        dto = oa.step()
        assert abs(dt - dto) <= 1e-8 * dto
        _compare(amr, oa, 2e-8, f"after coarse step {step + 1}")
        st, stm = amr.sync_stats()
        assert st.converged == 1 and stm.converged == 1
    # composite conservation of mass, tracer and momentum (periodic domain, conservative updates + reflux + sync)
    n = [oa.n(0), oa.n(1)]
    Sg = [amr.levels[l].data(0).gather_valid(n[l]) for l in range(2)]
    m1 = [_composite_sum(Sg, cov1, oa.dx(0), oa.dx(1), c) for c in range(ncomp)]
    assert abs(m1[3] - m0[3]) <= 1e-12
    if not case.startswith("two_tracers"):
        assert abs(m1[4] - m0[4]) <= 1e-11
    else:                                   # the conservatively advected tracer of the pair is conserved on the composite grid
        c = 4 if case.endswith("cons") else 5
        assert Sg[0].shape[-1] == 6 and abs(m1[c] - m0[c]) <= 1e-11



@pytest.mark.parametrize("case", ["viscous", "inviscid_boxes"])
def test_two_level_channel_refined_at_the_inflow_matches_oracle(case):
    """the refined level touches the inflow face (Exec/run2d regtest.2d.poiseuille: the tracer blob sits next to x-lo): the coarse and fine
    sync residuals count the velocity behind the inflow face only where the cell it mirrors counts (mlndlap_fillbc_cc of the cell masks
    in compSyncResidualCoarse / Fine); the coarse/fine corner on the face is where an error shows.  x: inflow / outflow, y: no-slip
    walls, z: periodic."""
    INFLOW, OUTFLOW, NOSLIP = 1, 2, 5
    wl = [0.0] * 9
    wl[0] = 1.0
    sl = [0.0] * 12
    sl[0], sl[1] = 1.0, 0.25                                     # xlo.density, xlo.tracer
    kw = dict(cfl=0.5, visc_coef=0.05, tracer_diff_coef=0.002, init_iter=2, init_shrink=0.3, phys_lo=[INFLOW, NOSLIP, 0],
              phys_hi=[OUTFLOW, NOSLIP, 0], wall_vel_lo=wl, scal_bc_lo=sl)
    fine = [([0, 8, 0], [15, 23, 31])]
    if case == "inviscid_boxes":                                  # two boxes along the face, one of them also on the lower wall
        kw.update(visc_coef=0.0, tracer_diff_coef=0.0)
        fine = [([0, 0, 0], [15, 15, 31]), ([0, 16, 0], [7, 23, 31])]

    def state(X, Y, Z):
        S = np.zeros(X.shape + (5,), order="F")
        S[..., 0] = 1.0
        S[..., 3] = 1.0
        S[..., 4] = np.exp(-((X - 0.15) ** 2 + (Y - 0.5) ** 2) / 0.01)
        return S
    amr, oa = _make(16, fine, 8, kw, state, periodic=(0, 0, 1))
    amr.post_init()
    oa.post_init()
    _compare(amr, oa, 2e-8, "after post_init")
    for step in range(2):
        dt = amr.coarse_step()
        dto = oa.step()
        assert abs(dt - dto) <= 1e-8 * dto
        _compare(amr, oa, 2e-8, f"after coarse step {step + 1}")
    S0 = amr.levels[0].data(0).gather_valid(oa.n(0))
    assert np.abs(S0[..., 0]).max() < 1.6                        # a wrong residual at the corner node shows as a jet along the face


def _composite_sum_n(states, covs, dxs, comp):
    tot = 0.0
    for l, S in enumerate(states):
        m = covs[l].copy()
        if l + 1 < len(states):
            m &= ~covs[l + 1][::2, ::2, ::2]
        tot += (S[..., comp] * m).sum() * np.prod(dxs[l])
    return tot


@pytest.mark.parametrize("case", ["nested_boxes", "ppm_gravity_walls", "viscous"])
def test_three_level_step_matches_oracle(case):
    """three levels: MAC sync and sync projection on a refined level (homogeneous coarse/fine data in mac_sync_solve, the level's own
    corrections entering the registers of the interface below, SyncRegister::CompAdd), SyncInterp over two levels (ratio 4),
    SyncProjInterp + computeGradP on the finest level, composite projections over three levels in post_init."""
    n0 = 8
    l1 = [([2, 2, 2], [13, 13, 13])]
    if case in ("nested_boxes", "viscous"):
        l2 = [([10, 10, 10], [21, 21, 21])]
        kw = dict(cfl=0.7, visc_coef=0.0, init_iter=2)
        if case == "viscous":
            # the viscous sync on a refined level: diffuse_Vsync / the scalar sync solve with homogeneous coarse/fine data, their fluxes into
            # the viscous register of the interface below (x dt^2 / x dt), viscous fluxes of level 1 as fine AND coarse side
            kw.update(visc_coef=0.01, tracer_diff_coef=0.005)
        fn = lambda X, Y, Z: orc.taylorgreen_state(X, Y, Z, c=1.0)
        per = (1, 1, 1)
    else:
        # the ingredients of Exec/run3d/inputs.3d.rt (regression test C5): PPM, conservative momentum and tracer, gravity, slip walls in z
        l2 = [([10, 10, 10], [15, 21, 21]), ([16, 10, 10], [21, 21, 21])]
        kw = dict(cfl=0.7, visc_coef=0.0, init_iter=2, use_ppm=1, do_mom_diff=1, do_cons_trac=1, gravity=-1.0,
                  phys_lo=[0, 0, 4], phys_hi=[0, 0, 4])
        per = (1, 1, 0)

        def fn(X, Y, Z):
            S = orc.taylorgreen_state(X, Y, Z, c=1.0)
            S[..., 2] *= np.sin(np.pi * Z)                       # no flow through the walls
            S[..., 3] = 1.0 + 0.5 * (1.0 + np.tanh((Z - 0.5 - 0.05 * np.cos(2 * np.pi * X) * np.cos(2 * np.pi * Y)) / 0.1))
            return S
    amr, oa = _make(n0, l1, 8, kw, fn, finer=[l2], periodic=per)
    amr.post_init()
    oa.post_init()
    for l in range(3):
        assert abs(amr.dts()[l] - oa.dt(l)) <= 1e-9 * oa.dt(l)
    _compare(amr, oa, 2e-8, "after post_init")
    covs = [oa.cov(l) for l in range(3)]
    dxs = [oa.dx(l) for l in range(3)]
    m0 = [_composite_sum_n([oa.state(l) for l in range(3)], covs, dxs, c) for c in range(5)]
    for step in range(2):
        dt = amr.coarse_step()
        dto = oa.step()
        assert abs(dt - dto) <= 1e-8 * dto
        _compare(amr, oa, 2e-8, f"after coarse step {step + 1}")
    Sg = [amr.levels[l].data(0).gather_valid(oa.n(l)) for l in range(3)]
    m1 = [_composite_sum_n(Sg, covs, dxs, c) for c in range(5)]
    assert abs(m1[3] - m0[3]) <= 1e-12
    assert abs(m1[4] - m0[4]) <= 1e-11


def test_improperly_nested_level_is_refused():
    """a level-2 box whose Godunov ghost cells + interpolation stencil reach outside level 1 (what amrex::Amr never generates)"""
    from iamr_amd import lib as L
    from iamr_amd.ns import ns_params
    from iamr_amd.amr import Amr
    L.init()
    lays = [L.Layout.single([8] * 3), L.Layout([((2, 2, 2), (13, 13, 13))]), L.Layout([((8, 8, 8), (15, 15, 15))])]
    with pytest.raises(L.IamrxError, match="properly nested"):
        Amr(L.Geom.make([8] * 3), lays, ns_params(), L.mg_opts())


def test_two_level_lid_driven_cavity_refined_at_the_lid():
    """viscous hierarchy with walls: the lid-driven cavity of config C4 (no-slip / slip walls, moving lid, tracer diffusion) with a refined
    box that touches the lid and two side walls -- coarse/fine faces of the tensor and scalar solves meeting physical boundaries,
    viscous flux registers, viscous sync.  Product vs oracle after the initial iterations and two coarse steps."""
    n0 = 16
    fine = [([0, 8, 16], [31, 23, 31])]            # x: wall to wall, z: up to the lid
    lid = [0.0] * 9
    lid[6] = 1.0
    kw = dict(cfl=0.3, visc_coef=0.01, tracer_diff_coef=0.001, init_iter=2, init_dt=0.0140625 * 4, init_shrink=0.3,
              phys_lo=[4, 4, 5], phys_hi=[5, 5, 5], wall_vel_hi=lid)

    def rest(X, Y, Z):
        S = np.zeros(X.shape + (5,), order="F")
        S[..., 3] = 1.0
        S[..., 4] = np.exp(-((X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.75) ** 2) / 0.02)
        return S
    amr, oa = _make(n0, fine, 8, kw, rest, periodic=(0, 0, 0))
    amr.post_init()
    oa.post_init()
    _compare(amr, oa, 2e-8, "after post_init")
    for step in range(2):
        dt = amr.coarse_step()
        dto = oa.step()
        assert abs(dt - dto) <= 1e-8 * dto
        _compare(amr, oa, 2e-8, f"after coarse step {step + 1}")
