"""BDS advection (ns.advection_scheme = BDS; Source/NavierStokesBase.cpp:548-553, 4701-4717) through the C-ABI against the oracle's
restatement (oracle/orc_bds.c, pinned on known answers by tests/test_cpu_bds.py): edge states and aofs of the raw ComputeAofs entry on
periodic and wall-bounded boxes (conservative and convective components, forcing), and full time steps of the level driver."""
import ctypes as C
import numpy as np
import pytest
from test_gpu_godunov import field, periodic_fab, to_dev

pytestmark = pytest.mark.gpu
BDS = 2


@pytest.fixture(autouse=True)
def _oracle_scheme(orc):
    yield
    orc.lib().orc_godunov_set_ppm(0)


def same(got, ref, tag, rel=1e-12):
    err = float(np.abs(got - ref).max())
    assert err <= rel * max(1.0, float(np.abs(ref).max())), (tag, err)


@pytest.mark.parametrize("n,boxes,ncomp,iconserv,force", [((16, 16, 16), None, 3, (0, 0, 0), True), ((24, 16, 32), 8, 2, (1, 0), False),
                                                          ((16, 16, 16), 8, 5, (0, 0, 0, 1, 1), True)])
def test_bds_compute_aofs_periodic(orc, gpu, n, boxes, ncomp, iconserv, force):
    lib, L = gpu, orc.lib()
    L.orc_godunov_set_ppm(BDS)
    g_o, g_d = orc.geom(n), lib.Geom.make(n)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    S = periodic_fab(orc, L, g_o, n, orc.CELL, 3, ncomp, 300)
    frc = periodic_fab(orc, L, g_o, n, orc.CELL, 1, ncomp, 400, amp=2.0) if force else None
    um_o = []
    for d in range(3):
        t = orc.face(d)
        f = orc.Fab(n, t, 1, 1)
        f.a[..., 0] = field(n, 1, 600 + d, 1.0, t)
        f.a[np.abs(f.a) < 0.02] = 0.0
        hi, lo = [slice(None)] * 3, [slice(None)] * 3
        hi[d], lo[d] = 1 + n[d], 1
        f.a[tuple(hi)] = f.a[tuple(lo)]
        L.orc_fill_periodic(f.ref(), C.byref(g_o), orc.i3(t))
        um_o.append(f)
    aofs_o = orc.Fab(n, orc.CELL, 0, 5)
    edge_o = [orc.Fab(n, orc.face(d), 0, ncomp) for d in range(3)]
    ic = (C.c_int * ncomp)(*iconserv)
    dt = 0.4 / max(n)
    L.orc_compute_aofs(C.byref(g_o), aofs_o.ref(), 0, S.ref(), ncomp, frc.ref() if force else None, None, orc.fabptrs(um_o), ic,
                       C.c_double(dt), orc.bcrecs(ncomp), 0, 0, orc.fabptrs(edge_o), None)
    S_d = to_dev(lib, lay, S, lib.CELL, 3)
    frc_d = to_dev(lib, lay, frc, lib.CELL, 1) if force else None
    um_d = [to_dev(lib, lay, um_o[d], lib.face(d), 1) for d in range(3)]
    aofs_d = lib.MultiFab(lay, lib.CELL, 5, 0)
    aofs_d.setval(0.0)
    edge_d = [lib.MultiFab(lay, lib.face(d), ncomp, 0) for d in range(3)]
    lib.godunov_compute_aofs(g_d, aofs_d, 0, S_d, ncomp, frc_d, None, um_d, iconserv, dt, None, 0, 0, edge=edge_d, scheme=BDS)
    for d in range(3):
        same(edge_d[d].gather_valid(n), edge_o[d].valid(n, orc.face(d)), ("edge", d))
    same(aofs_d.gather_valid(n)[..., :ncomp], aofs_o.valid(n)[..., :ncomp], "aofs")
    # a different scheme, not a relabelled Godunov
    L.orc_godunov_set_ppm(0)
    edge_p = [orc.Fab(n, orc.face(d), 0, ncomp) for d in range(3)]
    L.orc_compute_aofs(C.byref(g_o), aofs_o.ref(), 0, S.ref(), ncomp, frc.ref() if force else None, None, orc.fabptrs(um_o), ic,
                       C.c_double(dt), orc.bcrecs(ncomp), 0, 0, orc.fabptrs(edge_p), None)
    assert np.abs(edge_p[0].a - edge_o[0].a).max() > 1e-4


@pytest.mark.parametrize("walls", [False, True])
def test_bds_time_steps_match_oracle(orc, gpu, walls):
    """three steps of the level driver with ns.advection_scheme = BDS: velocity, density and tracer advected by BDS edge states, the
    velocity prediction by Godunov_PLM; periodic box, and a box with slip walls in z (ghost faces of u_mac extrapolated)"""
    from iamr_amd import ns as N
    lib, L = gpu, orc.lib()
    n = (16, 16, 16)
    per = (1, 1, 0) if walls else (1, 1, 1)
    kw = dict(cfl=0.7, visc_coef=0.001 if walls else 0.0, init_iter=2, use_ppm=BDS, do_cons_trac=1)
    if walls:
        kw.update(phys_lo=[0, 0, 4], phys_hi=[0, 0, 4])
    g_o = orc.geom(n, periodic=per)
    s = C.c_void_p(L.orc_ns_create(C.byref(g_o), C.byref(orc.ns_params(**kw)), C.byref(orc.mg_opts())))
    X, Y, Z = np.meshgrid(*[(np.arange(n[d]) + 0.5) / n[d] for d in range(3)], indexing="ij")
    S0 = orc.taylorgreen_state(X, Y, Z, c=1.0)
    if walls:
        S0[..., 2] *= np.sin(np.pi * Z)
    S0[..., 3] = 1.0 + 0.3 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y)
    S0[..., 4] = S0[..., 3] * np.exp(-((X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.5) ** 2) / 0.02)
    orc.from_cfab(L.orc_ns_fab(s, 0)).valid(n)[...] = S0
    L.orc_ns_post_init(s, C.c_double(-1.0))
    dts_o = [L.orc_ns_step(s) for _ in range(3)]
    ref = orc.from_cfab(L.orc_ns_fab(s, 0)).valid(n).copy()
    L.orc_ns_destroy(s)
    g_d = lib.Geom.make(n, periodic=per)
    lay = lib.Layout.decompose(n, 8)
    ns = N.NavierStokes(g_d, lay, N.ns_params(**kw))
    m = lib.MultiFab(lay, lib.CELL, 5, 1)
    G = np.zeros(tuple(v + 2 for v in n) + (5,), order="F")
    G[1:-1, 1:-1, 1:-1] = S0
    m.set_from_global(G, (-1, -1, -1))
    ns.set_data(ns.S_NEW, m)
    ns.post_init(-1.0)
    dts = [ns.step() for _ in range(3)]
    got = ns.data(ns.S_NEW).gather_valid(n)
    assert np.allclose(dts, dts_o, rtol=1e-9, atol=0)
    for c in range(5):
        assert np.abs(got[..., c] - ref[..., c]).max() <= 2e-8 * max(1.0, np.abs(ref[..., c]).max()), c
    # conservative components: mass and tracer mass are conserved by the BDS fluxes
    assert abs(got[..., 3].sum() - S0[..., 3].sum()) < 1e-10 * S0[..., 3].sum()
    assert abs(got[..., 4].sum() - S0[..., 4].sum()) < 1e-10 * S0[..., 4].sum()
