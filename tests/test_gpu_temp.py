"""ns.do_temp (Exec/run3d/regtest.3d.hotspot): temperature as the last state component and the divergence
constraint div U = S = div(lambda grad T) / (rho T) -- NavierStokes::calc_divu (NavierStokes.cpp:1876-1958), calc_dsdt, create_mac_rhs
(NavierStokesBase.cpp:818-858, 1038-1065), the cell-centred source of the nodal projections (Projection.cpp:267-276, 379-389, 732-788,
1008-1148) and of the sync residuals, the constraint in create_umac_grown (NavierStokesBase.cpp:1235) and in the advective updates.
HIP library against the CPU oracle on one level (one box / eight boxes) and on a two-level hierarchy, and the reference's own
hotspot inputs, unmodified."""
import ctypes as C
import os

import numpy as np
import pytest

import orc as orcmod

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
SLIP, OUTFLOW, NOSLIP = 4, 2, 5


def _hot_state(X, Y, Z, ncomp):
    S = np.zeros(X.shape + (ncomp,), order="F")
    rho = 1.0 / (1.0 + 0.5 * np.exp(-((X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.4) ** 2) / 0.02))
    S[..., 3] = rho
    S[..., 4] = np.exp(-((X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.4) ** 2) / 0.01)
    S[..., 5] = 1.0 / rho
    return S


@pytest.mark.parametrize("boxes", [None, 8])
def test_single_level_hot_bubble_matches_oracle(orc, gpu, boxes):
    from iamr_amd import ns as N
    L = orc.lib()
    n = (16, 16, 16)
    per = (1, 1, 0)
    kw = dict(cfl=0.5, visc_coef=0.01, init_iter=2, init_shrink=0.3, do_temp=1, temp_cond_coef=1.0e-3, tracer_diff_coef=0.002, gravity=-1.0)
    g = orc.geom(n, probhi=(1.0, 1.0, 1.0), periodic=per)
    p = orc.CNsParams()
    L.orc_ns_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    p.phys_lo[2], p.phys_hi[2] = SLIP, OUTFLOW
    o = orc.mg_opts()
    s = C.c_void_p(L.orc_ns_create(C.byref(g), C.byref(p), C.byref(o)))
    assert s.value
    x = (np.arange(16) + 0.5) / 16
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
    S0 = _hot_state(X, Y, Z, 6)
    L.orc_ns_init_rest(s, C.c_double(1.0))
    orc.from_cfab(L.orc_ns_fab(s, 0)).a[1:-1, 1:-1, 1:-1, :6] = S0
    L.orc_ns_post_init(s, C.c_double(-1.0))
    dts_o = [L.orc_ns_step(s) for _ in range(3)]
    S_o = orc.from_cfab(L.orc_ns_fab(s, 0)).valid(n).copy()
    P_o = orc.from_cfab(L.orc_ns_fab(s, 2)).valid(n, orc.NODE).copy()
    L.orc_ns_destroy(s)

    gg = gpu.Geom.make(n, prob_hi=(1.0, 1.0, 1.0), periodic=per)
    lay = gpu.Layout.decompose(n, boxes) if boxes else gpu.Layout.single(n)
    ns = N.NavierStokes(gg, lay, N.ns_params(phys_lo=[0, 0, SLIP], phys_hi=[0, 0, OUTFLOW], **kw))
    assert ns.nstate == 6 and ns.nalloc == 8
    ns.init_rest(1.0)
    m = gpu.MultiFab(lay, gpu.CELL, 6, 1)
    G = np.zeros(tuple(v + 2 for v in n) + (6,))
    G[1:-1, 1:-1, 1:-1] = S0
    m.set_from_global(G, (-1, -1, -1))
    ns.set_data(N.NavierStokes.S_NEW, m)
    ns.post_init(-1.0)
    dts = [ns.step() for _ in range(3)]
    assert np.allclose(dts, dts_o, rtol=1e-9, atol=0.0)
    S = ns.data(N.NavierStokes.S_NEW).gather_valid(n)
    assert S.shape[-1] == 8
    vs = np.abs(S_o[..., :3]).max()
    for comp in range(8):
        scale = vs if comp < 3 else max(np.abs(S_o[..., comp]).max(), 1e-3)
        assert np.abs(S[..., comp] - S_o[..., comp]).max() <= 5e-8 * scale, comp
    assert np.abs(S_o[..., 6]).max() > 1e-3                                          # the constraint is active
    Pd = ns.data(N.NavierStokes.P_NEW).gather_valid(n)[..., 0]
    assert np.abs(Pd - P_o[..., 0]).max() <= 1e-6 * np.abs(P_o).max()


def test_two_level_hot_bubble_matches_oracle(gpu):
    """the refined level sits on the bubble: coarse/fine divergence fix of the MAC velocity with the constraint, sync residuals with the
    cell-centred source, divu / dsdt averaged down and interpolated like the state"""
    from iamr_amd import lib as L
    from iamr_amd.ns import ns_params
    from iamr_amd.amr import Amr
    n0 = 16
    per = (1, 1, 0)
    kw = dict(cfl=0.5, visc_coef=0.01, init_iter=2, init_shrink=0.3, do_temp=1, temp_cond_coef=1.0e-3, gravity=-1.0,
              phys_lo=[0, 0, SLIP], phys_hi=[0, 0, NOSLIP])
    fine = [([8, 8, 4], [23, 23, 19])]
    g0 = L.Geom.make([n0] * 3, periodic=per)
    lays = [L.Layout.decompose([n0] * 3, 8), L.Layout([(tuple(lo), tuple(hi)) for lo, hi in fine])]
    amr = Amr(g0, lays, ns_params(**kw), L.mg_opts())
    og = orcmod.geom([n0] * 3, periodic=per)
    oa = orcmod.OrcAmr(og, orcmod.ns_params(**kw), orcmod.mg_opts(), [[], fine])
    for l in range(2):
        S = _hot_state(*oa.cell_centres(l), 6)
        oa.fab(l, 0).valid(oa.n(l))[..., :6] = S
        lev = amr.levels[l]
        mf = L.MultiFab(lev.layout, L.CELL, 6, 1)
        G = np.zeros(tuple(v + 2 for v in S.shape[:3]) + (6,), order="F")
        G[1:-1, 1:-1, 1:-1] = S
        mf.set_from_global(G, (-1, -1, -1))
        lev.set_data(lev.S_NEW, mf)
    amr.post_init()
    oa.post_init()
    assert abs(amr.dts()[0] - oa.dt(0)) <= 1e-9 * oa.dt(0)
    for step in range(2):
        dt = amr.coarse_step()
        dto = oa.step()
        assert abs(dt - dto) <= 1e-8 * dto
        for l in range(2):
            n = oa.n(l)
            cov = oa.cov(l)
            S = amr.levels[l].data(0).gather_valid(n)
            So = oa.state(l)
            vs = np.abs(So[..., :3][cov]).max()
            for comp in range(8):
                scale = vs if comp < 3 else max(np.abs(So[..., comp][cov]).max(), 1e-3)
                err = np.abs(S[..., comp] - So[..., comp])[cov].max()
                assert err <= 1e-7 * scale, (step, l, comp, err, scale)
    assert np.abs(oa.state(1)[..., 6]).max() > 1e-3


def test_reference_hotspot_regtest_inputs(gpu, tmp_path, capsys):
    """Exec/run3d/regtest.3d.hotspot, unmodified: a hot, light bubble rising under gravity towards an outflow boundary; do_temp with
    temp_cond_coef = 1e-8, a second (conservative) tracer, walls on the sides, two refined levels tagged on the temperature and its
    differences, ns.do_refine_outflow = 1, regrid every second step"""
    from iamr_amd import run as R
    from iamr_amd.plotfile import PlotFile
    root = str(tmp_path / "plt")
    assert R.main([os.path.join(HERE, "golden", "regtest.3d.hotspot"), "max_step=4", "amr.plot_int=4", f"amr.plot_file={root}"]) == 0
    out = capsys.readouterr().out
    steps = [l for l in out.splitlines() if l.startswith("STEP =")]
    assert len(steps) == 4, out[-2000:]
    pf = PlotFile.read(root + "00004")
    assert pf.names[5:] == ["tracer2", "temp", "divu", "dsdt"] and len(pf.levels) >= 2
    for lv in pf.levels:
        for a in lv.data:
            assert np.isfinite(a).all() and a[..., 3].min() > 0.4 and a[..., 6].max() < 2.1
    w = max(a[..., 2].max() for a in pf.levels[0].data)
    assert w > 0.0                                                                   # the bubble has started to rise
