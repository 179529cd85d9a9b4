"""Godunov_PPM (ns.advection_scheme, reference Source/NavierStokesBase.cpp:548-553, 4654-4656; config C5's
Exec/run3d/regtest.3d.rayleightaylor:8).  HIP kernels with the PPM switch on against the CPU oracle's restatement of the
Colella-Woodward reconstruction (oracle/orc_godunov.c): kernel parity to 1e-13 (bit-exact for the STRICT_FP build) incl. ext_dir /
extrapolation walls, and the RayleighTaylor regtest's physics (PPM, do_mom_diff, do_cons_trac, forces in the transverse terms)."""
import ctypes as C
import numpy as np
import pytest
from conftest import godunov_same
from test_gpu_godunov import field, periodic_fab, to_dev

pytestmark = pytest.mark.gpu


@pytest.fixture()
def ppm(orc, gpu):
    # the oracle keeps a global switch (test infrastructure); the product takes the scheme as an argument of every call
    orc.lib().orc_godunov_set_ppm(1)
    yield
    orc.lib().orc_godunov_set_ppm(0)


@pytest.mark.parametrize("n,boxes,fit", [((16, 16, 16), None, 0), ((32, 16, 24), 8, 1)])
def test_ppm_extrap_vel_to_faces(orc, gpu, ppm, n, boxes, fit):
    lib, L = gpu, orc.lib()
    g_o, g_d = orc.geom(n), lib.Geom.make(n)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    vel = periodic_fab(orc, L, g_o, n, orc.CELL, 3, 3, 100)
    vel.a[np.abs(vel.a) < 0.02] = 0.0
    L.orc_fill_periodic(vel.ref(), C.byref(g_o), orc.i3(orc.CELL))
    force = periodic_fab(orc, L, g_o, n, orc.CELL, 1, 3, 200, amp=3.0)
    um_o = [orc.Fab(n, orc.face(d), 1, 1) for d in range(3)]
    dt = 0.4 / max(n)
    L.orc_extrap_vel_to_faces(C.byref(g_o), vel.ref(), force.ref(), orc.fabptrs(um_o), C.c_double(dt), orc.bcrecs(3), fit)
    um_plm = [orc.Fab(n, orc.face(d), 1, 1) for d in range(3)]
    L.orc_godunov_set_ppm(0)
    L.orc_extrap_vel_to_faces(C.byref(g_o), vel.ref(), force.ref(), orc.fabptrs(um_plm), C.c_double(dt), orc.bcrecs(3), fit)
    L.orc_godunov_set_ppm(1)
    vel_d, force_d = to_dev(lib, lay, vel, lib.CELL, 3), to_dev(lib, lay, force, lib.CELL, 1)
    um_d = [lib.MultiFab(lay, lib.face(d), 1, 1) for d in range(3)]
    lib.godunov_extrap_vel_to_faces(g_d, vel_d, force_d, um_d, dt, None, fit, scheme=1)
    for d in range(3):
        got, ref = um_d[d].gather_valid(n)[..., 0], um_o[d].valid(n, orc.face(d))[..., 0]
        godunov_same(got, ref, d)
        assert np.abs(ref - um_plm[d].valid(n, orc.face(d))[..., 0]).max() > 1e-4       # PPM is a different scheme, not a relabelled PLM


@pytest.mark.parametrize("n,boxes,ncomp,iconserv,isvel,fit", [((16, 16, 16), None, 3, (0, 0, 0), 1, 0), ((32, 16, 24), 8, 2, (1, 0), 0, 1)])
def test_ppm_compute_aofs(orc, gpu, ppm, n, boxes, ncomp, iconserv, isvel, fit):
    lib, L = gpu, orc.lib()
    g_o, g_d = orc.geom(n), lib.Geom.make(n)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    S = periodic_fab(orc, L, g_o, n, orc.CELL, 3, ncomp, 300)
    force = periodic_fab(orc, L, g_o, n, orc.CELL, 1, ncomp, 400, amp=2.0)
    divu = periodic_fab(orc, L, g_o, n, orc.CELL, 1, 1, 500, amp=0.3)
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
    L.orc_compute_aofs(C.byref(g_o), aofs_o.ref(), 1, S.ref(), ncomp, force.ref(), divu.ref(), orc.fabptrs(um_o), ic,
                       C.c_double(dt), orc.bcrecs(ncomp), isvel, fit, orc.fabptrs(edge_o), None)
    S_d, force_d, divu_d = to_dev(lib, lay, S, lib.CELL, 3), to_dev(lib, lay, force, lib.CELL, 1), to_dev(lib, lay, divu, lib.CELL, 1)
    um_d = [to_dev(lib, lay, um_o[d], lib.face(d), 1) for d in range(3)]
    aofs_d = lib.MultiFab(lay, lib.CELL, 5, 0)
    aofs_d.setval(0.0)
    edge_d = [lib.MultiFab(lay, lib.face(d), ncomp, 0) for d in range(3)]
    lib.godunov_compute_aofs(g_d, aofs_d, 1, S_d, ncomp, force_d, divu_d, um_d, iconserv, dt, None, isvel, fit, edge=edge_d, scheme=1)
    for d in range(3):
        godunov_same(edge_d[d].gather_valid(n), edge_o[d].valid(n, orc.face(d)), ("edge", d))
    godunov_same(aofs_d.gather_valid(n)[..., 1:1 + ncomp], aofs_o.valid(n)[..., 1:1 + ncomp], "aofs")


def test_rayleigh_taylor_regtest_physics_with_ppm(orc, gpu):
    """Exec/run3d/regtest.3d.rayleightaylor:6-8 (single level): Godunov_PPM + do_mom_diff + do_cons_trac + use_forces_in_trans,
    gravity, slip walls in z -- three steps, HIP level driver vs oracle"""
    from iamr_amd import ns as N
    lib, L = gpu, orc.lib()
    n, per, prob_lo, prob_hi = (16, 16, 32), (1, 1, 0), (0.0, 0.0, 0.0), (1.0, 1.0, 2.0)
    kw = dict(cfl=0.7, visc_coef=0.0, init_iter=3, gravity=-9.8, use_forces_in_trans=1, do_mom_diff=1, do_cons_trac=1, use_ppm=1,
              phys_lo=[0, 0, 4], phys_hi=[0, 0, 4])
    rt = dict(rho_1=1.0, rho_2=2.0, tra_1=1.0, tra_2=0.0, pertamp=0.1, interface_width=0.08)
    g_o = orc.geom(n, problo=prob_lo, probhi=prob_hi, periodic=per)
    s = C.c_void_p(L.orc_ns_create(C.byref(g_o), C.byref(orc.ns_params(**kw)), C.byref(orc.mg_opts())))
    L.orc_ns_init_rayleightaylor(s, *[C.c_double(rt[k]) for k in ("rho_1", "rho_2", "tra_1", "tra_2", "pertamp", "interface_width")])
    L.orc_ns_post_init(s, C.c_double(-1.0))
    dts_o = [L.orc_ns_step(s) for _ in range(3)]
    S_o = orc.from_cfab(L.orc_ns_fab(s, 0)).valid(n).copy()
    L.orc_ns_destroy(s)
    L.orc_godunov_set_ppm(0)
    g_d = lib.Geom.make(n, prob_lo=prob_lo, prob_hi=prob_hi, periodic=per)
    ns = N.NavierStokes(g_d, lib.Layout.decompose(n, 16), N.ns_params(**kw))
    ns.init_rayleightaylor(**rt)
    ns.post_init(-1.0)
    dts = [ns.step() for _ in range(3)]
    assert np.allclose(dts, dts_o, rtol=1e-9, atol=0)
    S = ns.data(N.NavierStokes.S_NEW).gather_valid(n)
    for comp in range(5):
        scale = max(np.abs(S_o[..., comp]).max(), 1e-3)
        assert np.abs(S[..., comp] - S_o[..., comp]).max() <= 1e-8 * scale, comp
    assert np.abs(S[..., 2]).max() > 1e-3


@pytest.mark.parametrize("boxes", [None, 8])
def test_ppm_with_walls(orc, gpu, ppm, boxes):
    """the wall test of the PLM kernels (no-slip / slip walls, moving lid: ext_dir, hoextrap and foextrap faces) with the PPM switch on:
    one-sided edge values next to ext_dir / hoextrap faces"""
    from test_gpu_walls import test_godunov_with_walls
    test_godunov_with_walls(orc, gpu, boxes, scheme=1)
