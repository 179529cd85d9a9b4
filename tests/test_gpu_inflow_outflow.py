"""Inflow / outflow boundaries (NS_BC.H tables, MacProj::set_mac_solve_bc, Projection::set_boundary_velocity and
the LinOpBCType::inflow / Dirichlet nodal BCs of Projection::doMLMGNodalProjection): channel flow driven by a uniform inflow,
product against the CPU oracle, plus the two pins the physics offers: the volume flux through every cross-section equals the
inflow flux (discrete incompressibility), and far from the inlet the profile approaches plane Poiseuille flow."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
INFLOW, OUTFLOW, SLIP, NOSLIP = 1, 2, 4, 5


def params(n, prob_hi, per, lo, hi, uin, rho_in, trac_in, **kw):
    wl = [0.0] * 9
    wl[0] = uin                      # xlo.velocity = uin 0 0
    sl = [0.0] * 6
    sl[0], sl[1] = rho_in, trac_in   # xlo.density, xlo.tracer
    return dict(phys_lo=lo, phys_hi=hi, wall_vel_lo=wl, scal_bc_lo=sl, **kw)


def run_oracle(orc, n, prob_hi, per, P, nsteps, init_u):
    L = orc.lib()
    g = orc.geom(n, probhi=prob_hi, periodic=per)
    p = orc.CNsParams()
    L.orc_ns_default_params(C.byref(p))
    for k, v in P.items():
        if k in ("phys_lo", "phys_hi"):
            for d in range(3):
                getattr(p, k)[d] = 0 if per[d] else v[d]
        elif k in ("wall_vel_lo", "scal_bc_lo"):
            for q, x in enumerate(v):
                getattr(p, k)[q] = x
        else:
            setattr(p, k, v)
    o = orc.mg_opts()
    s = C.c_void_p(L.orc_ns_create(C.byref(g), C.byref(p), C.byref(o)))
    assert s.value
    L.orc_ns_init_rest(s, C.c_double(1.0))
    orc.from_cfab(L.orc_ns_fab(s, 0)).a[1:-1, 1:-1, 1:-1, 0] = init_u
    L.orc_ns_post_init(s, C.c_double(-1.0))
    dts = [L.orc_ns_step(s) for _ in range(nsteps)]
    S = orc.from_cfab(L.orc_ns_fab(s, 0)).valid(n).copy()
    Pn = orc.from_cfab(L.orc_ns_fab(s, 2)).valid(n, orc.NODE).copy()
    T = L.orc_ns_time(s)
    L.orc_ns_destroy(s)
    return S, Pn, T, dts


def run_gpu(lib, n, prob_hi, per, P, nsteps, init_u, boxes):
    from iamr_amd import ns as N
    g = lib.Geom.make(n, prob_hi=prob_hi, periodic=per)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    P = dict(P)
    P["phys_lo"] = [0 if per[d] else P["phys_lo"][d] for d in range(3)]
    P["phys_hi"] = [0 if per[d] else P["phys_hi"][d] for d in range(3)]
    ns = N.NavierStokes(g, lay, N.ns_params(**P))
    ns.init_rest(1.0)
    m = lib.MultiFab(lay, lib.CELL, 5, 1)
    G = np.zeros(tuple(v + 2 for v in n) + (5,))
    G[1:-1, 1:-1, 1:-1, 0] = init_u
    G[1:-1, 1:-1, 1:-1, 3] = 1.0
    m.set_from_global(G, (-1, -1, -1))
    ns.set_data(N.NavierStokes.S_NEW, m)
    ns.post_init(-1.0)
    dts = [ns.step() for _ in range(nsteps)]
    return ns, lay, g, dts


@pytest.mark.parametrize("boxes", [None, 8])
def test_channel_matches_oracle(orc, gpu, boxes):
    """x: inflow (u = 1, rho = 1, tracer = 0.5) / outflow, y: no-slip walls, z: slip walls; 2 pressure iterations + 4 steps"""
    from iamr_amd import ns as N
    n, prob_hi, per = (32, 16, 8), (2.0, 1.0, 0.5), (0, 0, 0)
    P = params(n, prob_hi, per, (INFLOW, NOSLIP, SLIP), (OUTFLOW, NOSLIP, SLIP), 1.0, 1.0, 0.5,
               cfl=0.5, visc_coef=0.05, init_iter=2, init_shrink=0.3, tracer_diff_coef=0.002)
    S_o, P_o, T_o, dts_o = run_oracle(orc, n, prob_hi, per, P, 4, 1.0)
    ns, lay, g, dts = run_gpu(gpu, n, prob_hi, per, P, 4, 1.0, boxes)
    assert np.allclose(dts, dts_o, rtol=1e-9, atol=0.0) and abs(ns.time - T_o) <= 1e-12
    S = ns.data(N.NavierStokes.S_NEW).gather_valid(n)
    for comp in range(5):
        scale = max(np.abs(S_o[..., comp]).max(), 1e-3)
        assert np.abs(S[..., comp] - S_o[..., comp]).max() <= 1e-8 * scale, comp
    Pd = ns.data(N.NavierStokes.P_NEW).gather_valid(n)[..., 0]
    assert np.abs(Pd - P_o[..., 0]).max() <= 1e-6 * max(np.abs(P_o).max(), 1e-3)      # no free constant: Dirichlet at the outflow
    assert np.abs(Pd[-1]).max() == 0.0                                               # outflow nodes hold p = 0
    # discrete incompressibility of the MAC field, and the flux through every cross-section is the inflow flux
    um = [ns.data(6 + d) for d in range(3)]
    div = gpu.MultiFab(lay, gpu.CELL, 1, 0)
    gpu.mac_divergence(g, div, um)
    assert div.norm0() <= 1e-9
    ux = um[0].gather_valid(n)[..., 0] if hasattr(um[0], "gather_valid") else None
    if ux is not None:
        flux = ux.mean(axis=(1, 2))
        assert np.abs(flux - 1.0).max() <= 1e-10
    assert S[..., 4].max() > 0.3 and S[..., 4].max() <= 0.5 + 1e-9                   # the inflow tracer value enters the domain


def test_channel_develops_plane_poiseuille_flow(gpu):
    """Long channel (L = 4 H), Re = U H / nu = 10, z periodic: at the outlet the velocity profile is the plane Poiseuille
    parabola u(y) = 6 U y (1 - y) (cell centres), to the accuracy of a 16-cell second-order discretisation"""
    from iamr_amd import ns as N
    n, prob_hi, per = (64, 16, 4), (4.0, 1.0, 0.25), (0, 0, 1)
    P = params(n, prob_hi, per, (INFLOW, NOSLIP, 0), (OUTFLOW, NOSLIP, 0), 1.0, 1.0, 0.0,
               cfl=0.7, visc_coef=0.1, init_iter=2, init_shrink=0.3)
    ns, lay, g, dts = run_gpu(gpu, n, prob_hi, per, P, 0, 1.0, 16)
    while ns.time < 3.0:
        ns.step()
    S = ns.data(N.NavierStokes.S_NEW).gather_valid(n)
    u = S[-4, :, 0, 0]
    y = (np.arange(n[1]) + 0.5) / n[1]
    exact = 6.0 * y * (1.0 - y)
    assert np.abs(u - exact).max() <= 0.02, np.abs(u - exact).max()
    assert np.abs(S[-4, :, :, 1]).max() <= 5e-3                                      # no cross flow left
    assert abs(S[:, :, :, 0].mean(axis=(1, 2)) - 1.0).max() <= 1e-6                  # cell-centred flux conserved too
