"""GPU parity: nodal projection, tensor diffusion and the full NavierStokes::advance sequence (HIP, through
the C-ABI) against the CPU oracle, plus analytic (Taylor vortex) and invariant checks on the GPU result."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def cc(n, ng):
    return [(np.arange(-ng, n[d] + ng) + 0.5) / n[d] for d in range(3)]


def test_nodal_operator_and_projection(orc, gpu):
    lib = gpu
    from iamr_amd import ns as N
    L = orc.lib()
    n = (16, 16, 16)
    g_o = orc.geom(n)
    g_d = lib.Geom.make(n)
    lay = lib.Layout.single(n)
    X, Y, Z = np.meshgrid(*cc(n, 1), indexing="ij")
    sig = orc.Fab(n, orc.CELL, 1, 1)
    sig.a[..., 0] = 1.0 / (1.0 + 0.5 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) + 0.2 * np.cos(4 * np.pi * Z))
    vel = orc.Fab(n, orc.CELL, 1, 3)
    vel.a[..., 0] = np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) + 0.3 * np.cos(2 * np.pi * Z) * np.sin(4 * np.pi * X)
    vel.a[..., 1] = -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y) + 0.2 * np.sin(2 * np.pi * Y)
    vel.a[..., 2] = 0.5 * np.sin(2 * np.pi * Z) * np.cos(2 * np.pi * X)
    xn = [np.arange(-1, n[d] + 2) / n[d] for d in range(3)]
    Xn, Yn, Zn = np.meshgrid(*xn, indexing="ij")
    phi = orc.Fab(n, orc.NODE, 1, 1)
    phi.a[..., 0] = np.cos(2 * np.pi * (Xn + 2 * Yn)) * np.sin(2 * np.pi * Zn) + 0.1 * np.sin(6 * np.pi * Xn)
    # operator: explicit 27-point HIP stencil vs element-by-element oracle assembly (tolerance: rounding only)
    y = orc.Fab(n, orc.NODE, 0, 1)
    L.orc_nodal_adotx(C.byref(g_o), y.ref(), phi.ref(), sig.ref())
    sig_d = lib.MultiFab(lay, lib.CELL, 1, 1); sig_d.set_from_global(sig.a, sig.lo)
    phi_d = lib.MultiFab(lay, lib.NODE, 1, 1); phi_d.set_from_global(phi.a, phi.lo)
    out_d = lib.MultiFab(lay, lib.NODE, 1, 0)
    N.nodal_residual(g_d, out_d, phi_d, sig_d, None)
    got = out_d.gather_valid(n)[..., 0]
    assert np.abs(got - y.a[..., 0]).max() <= 1e-12 * np.abs(y.a).max()
    # divu rhs: same summation order -> bit exact
    r = orc.Fab(n, orc.NODE, 0, 1)
    L.orc_nodal_divu(C.byref(g_o), r.ref(), vel.ref())
    vel_d = lib.MultiFab(lay, lib.CELL, 3, 1); vel_d.set_from_global(vel.a, vel.lo)
    rhs_d = lib.MultiFab(lay, lib.NODE, 1, 0)
    N.nodal_divu(g_d, rhs_d, vel_d, 0)
    assert np.array_equal(rhs_d.gather_valid(n)[..., 0], r.a[..., 0])
    # full projection
    z3 = orc.i3([0, 0, 0])
    p_o = orc.Fab(n, orc.NODE, 1, 1)
    v_o = vel.copy()
    st_o = orc.CMgStats()
    oo = orc.mg_opts()
    L.orc_nodal_project(C.byref(g_o), v_o.ref(), p_o.ref(), sig.ref(), z3, z3, C.c_double(1e-12), C.c_double(1e-16), C.byref(oo), C.byref(st_o))
    p_d = lib.MultiFab(lay, lib.NODE, 1, 1); p_d.setval(0.0)
    gp_d = lib.MultiFab(lay, lib.CELL, 3, 1)
    st = N.nodal_projection(g_d, vel_d, 0, p_d, sig_d, 0, gp=gp_d, opts=lib.mg_opts(**orc.UPSTREAM_NODAL_CYCLE))
    assert st.converged == 1 and st.iters == st_o.iters
    pg = p_d.gather_valid(n)[..., 0]; pr = p_o.valid(n, orc.NODE)[..., 0]
    assert np.abs((pg - pg.mean()) - (pr - pr.mean())).max() <= 1e-9 * np.abs(pr).max()
    # the default hierarchy ends at 8^3 with the single-workgroup device bottom solver (k_nodal_bottom): same projection
    vel2_d = lib.MultiFab(lay, lib.CELL, 3, 1); vel2_d.set_from_global(vel.a, vel.lo)
    p2_d = lib.MultiFab(lay, lib.NODE, 1, 1); p2_d.setval(0.0)
    kw = dict(orc.UPSTREAM_NODAL_CYCLE); kw["device_bottom"] = 1
    st2 = N.nodal_projection(g_d, vel2_d, 0, p2_d, sig_d, 0, opts=lib.mg_opts(**kw))
    assert st2.converged == 1 and st2.iters <= st_o.iters + 1 and (st2.nlevels < st.nlevels or n[0] <= 8)
    p2 = p2_d.gather_valid(n)[..., 0]
    assert np.abs((p2 - p2.mean()) - (pr - pr.mean())).max() <= 1e-9 * np.abs(pr).max()
    assert np.abs(vel2_d.gather_valid(n) - v_o.valid(n)).max() <= 1e-9
    vg = vel_d.gather_valid(n); vr = v_o.valid(n)
    assert np.abs(vg - vr).max() <= 1e-9 * np.abs(vr).max()
    # Gp = grad(phi) (compGrad) consistent with the velocity update: vel_new = vel - sig * Gp
    gpg = gp_d.gather_valid(n)
    assert np.abs((vel.valid(n) - sig.valid(n) * gpg) - vg).max() <= 1e-12


def test_tensor_apply_and_solve(orc, gpu):
    lib = gpu
    from iamr_amd import ns as N
    L = orc.lib()
    n = (16, 16, 16)
    g_o = orc.geom(n)
    g_d = lib.Geom.make(n)
    lay = lib.Layout.single(n)
    X, Y, Z = np.meshgrid(*cc(n, 1), indexing="ij")
    u = orc.Fab(n, orc.CELL, 1, 3)
    u.a[..., 0] = np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) * np.cos(2 * np.pi * Z)
    u.a[..., 1] = np.cos(4 * np.pi * X) * np.sin(2 * np.pi * Y) + 0.3 * np.sin(2 * np.pi * Z)
    u.a[..., 2] = 0.5 * np.sin(2 * np.pi * (X + Y + Z))
    eta_o, eta_d = [], []
    for d in range(3):
        t = orc.face(d)
        ax = [(np.arange(0, n[q] + t[q]) + (0.0 if t[q] else 0.5)) / n[q] for q in range(3)]
        Xf, Yf, Zf = np.meshgrid(*ax, indexing="ij")
        e = orc.Fab(n, t, 0, 1)
        e.a[..., 0] = 0.01 * (1.0 + 0.3 * np.sin(2 * np.pi * Xf) * np.cos(2 * np.pi * Yf) * np.cos(2 * np.pi * Zf))
        eta_o.append(e)
        m = lib.MultiFab(lay, t, 1, 0); m.set_from_global(e.a, e.lo); eta_d.append(m)
    acoef = orc.Fab(n, orc.CELL, 0, 1)
    acoef.a[..., 0] = 1.0 + 0.2 * np.cos(2 * np.pi * X[1:-1, 1:-1, 1:-1])
    y = orc.Fab(n, orc.CELL, 0, 3)
    L.orc_tensor_apply(C.byref(g_o), y.ref(), u.ref(), C.c_double(0.0), C.c_double(-1.0), None, orc.fabptrs(eta_o))
    u_d = lib.MultiFab(lay, lib.CELL, 3, 1); u_d.set_from_global(u.a, u.lo)
    out_d = lib.MultiFab(lay, lib.CELL, 3, 0)
    N.tensor_apply(g_d, out_d, u_d, 0.0, -1.0, None, eta_d)
    got = out_d.gather_valid(n)
    assert np.abs(got - y.a).max() <= 1e-12 * np.abs(y.a).max()
    # implicit solve (1*acoef - 0.05 div tau) u = rhs
    rhs = orc.Fab(n, orc.CELL, 0, 3)
    rhs.a[...] = u.valid(n) * acoef.a
    z3 = orc.i3([0, 0, 0])
    s_o = u.copy()
    st_o = orc.CMgStats()
    oo = orc.mg_opts(maxorder=2)
    L.orc_tensor_solve(C.byref(g_o), s_o.ref(), rhs.ref(), C.c_double(1.0), C.c_double(0.05), acoef.ref(), orc.fabptrs(eta_o), z3, z3,
                       C.c_double(1e-10), C.c_double(0.0), C.byref(oo), C.byref(st_o))
    a_d = lib.MultiFab(lay, lib.CELL, 1, 0); a_d.set_from_global(acoef.a, acoef.lo)
    r_d = lib.MultiFab(lay, lib.CELL, 3, 0); r_d.set_from_global(rhs.a, rhs.lo)
    s_d = lib.MultiFab(lay, lib.CELL, 3, 1); s_d.set_from_global(u.a, u.lo)
    st = N.tensor_solve(g_d, s_d, r_d, 1.0, 0.05, a_d, eta_d, tol_rel=1e-10, tol_abs=0.0)
    assert st.converged == 1 and st.iters == st_o.iters
    assert np.abs(s_d.gather_valid(n) - s_o.valid(n)).max() <= 1e-8 * np.abs(s_o.valid(n)).max()


def run_oracle_tg(orc, N_, nsteps, visc, c, cfl=0.5):
    L = orc.lib()
    n = (N_,) * 3
    g = orc.geom(n)
    p = orc.CNsParams()
    L.orc_ns_default_params(C.byref(p))
    p.cfl = cfl; p.visc_coef = visc; p.init_iter = 2
    o = orc.mg_opts()
    s = C.c_void_p(L.orc_ns_create(C.byref(g), C.byref(p), C.byref(o)))
    L.orc_ns_init_taylorgreen(s, C.c_double(1.0), C.c_double(1.0), C.c_double(1.0), C.c_double(c), C.c_double(1.0))
    L.orc_ns_post_init(s, C.c_double(-1.0))
    dts = [L.orc_ns_step(s) for _ in range(nsteps)]
    S = orc.from_cfab(L.orc_ns_fab(s, 0)).valid(n).copy()
    Pn = orc.from_cfab(L.orc_ns_fab(s, 2)).valid(n, orc.NODE).copy()
    T = L.orc_ns_time(s)
    L.orc_ns_destroy(s)
    return S, Pn, T, dts


@pytest.mark.parametrize("boxes,c", [(None, 0.0), (8, 1.0)])
def test_advance_matches_oracle(orc, gpu, boxes, c):
    """TaylorGreen 16^3, init iterations + 3 steps: state and pressure vs the CPU oracle.
    Tolerance 1e-8 relative: the solvers stop at rtol 1e-12/1e-10 and reductions differ in order."""
    lib = gpu
    from iamr_amd import ns as N
    N_ = 16
    n = (N_,) * 3
    nsteps = 3
    visc = 1e-2
    S_o, P_o, T_o, dts_o = run_oracle_tg(orc, N_, nsteps, visc, c)
    g = lib.Geom.make(n)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.5, visc_coef=visc, init_iter=2))
    ns.init_taylorgreen(1.0, 1.0, 1.0, c, 1.0)
    ns.post_init(-1.0)
    dts = [ns.step() for _ in range(nsteps)]
    assert np.allclose(dts, dts_o, rtol=1e-9, atol=0.0)
    assert abs(ns.time - T_o) <= 1e-12
    S = ns.data(N.NavierStokes.S_NEW).gather_valid(n)
    for comp in range(5):
        scale = max(np.abs(S_o[..., comp]).max(), 1e-3)
        assert np.abs(S[..., comp] - S_o[..., comp]).max() <= 1e-8 * scale, comp
    Pd = ns.data(N.NavierStokes.P_NEW).gather_valid(n)[..., 0]
    Pr = P_o[..., 0]
    assert np.abs((Pd - Pd.mean()) - (Pr - Pr.mean())).max() <= 1e-6 * max(np.abs(Pr - Pr.mean()).max(), 1e-3)
    # invariants on the GPU result: density is conserved to round-off; MAC velocity is divergence free
    rho = S[..., 3]
    assert abs(rho.sum() - rho.size) <= 1e-9 * rho.size
    um = [ns.data(6 + d) for d in range(3)]
    div = lib.MultiFab(lay, lib.CELL, 1, 0)
    lib.mac_divergence(g, div, um)
    assert div.norm0() <= 1e-9


@pytest.mark.parametrize("boxes", [None, (64, 32, 32)])
def test_advance_at_config_c1_size_matches_oracle(orc, gpu, boxes):
    """BASELINE config C1 / the bench's workload at the size the oracle still finishes in seconds: TaylorGreen 64^3 (prob.c = 1; nu = 1e-4,
    cfl 0.7, init_iter 2: Tutorials/TaylorGreen/inputs.3d.taylorgreen), post_init + 2 steps against orc_ns_step.  At 64 cells per side the
    step runs the kernels that carry the 256^3 headline and that the 16^3 parity runs never select: the
    register-resident nodal pass k_nodal_gsr (boxes >= 48 cells in x and y), the pair-marching residual / restriction kernels, multi-tile
    k_god_z<14,14> / k_pred_z, the fused tensor residual; on four boxes kept as boxes (64 x 32 x 32) the ghost-filled nodal pass and the
    agglomerated coarse levels.  Same tolerances as the 16^3 test (solver tolerances 1e-12 / 1e-10, sums in another order)."""
    lib = gpu
    from iamr_amd import ns as N
    N_ = 64
    n = (N_,) * 3
    nsteps = 2
    visc = 1e-4
    S_o, P_o, T_o, dts_o = run_oracle_tg(orc, N_, nsteps, visc, 1.0, cfl=0.7)
    g = lib.Geom.make(n)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.7, visc_coef=visc, init_iter=2))
    ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
    ns.post_init(-1.0)
    dts = [ns.step() for _ in range(nsteps)]
    assert np.allclose(dts, dts_o, rtol=1e-9, atol=0.0)
    assert abs(ns.time - T_o) <= 1e-12
    S = ns.data(N.NavierStokes.S_NEW).gather_valid(n)
    for comp in range(5):
        scale = max(np.abs(S_o[..., comp]).max(), 1e-3)
        assert np.abs(S[..., comp] - S_o[..., comp]).max() <= 1e-8 * scale, (comp, float(np.abs(S[..., comp] - S_o[..., comp]).max()))
    Pd = ns.data(N.NavierStokes.P_NEW).gather_valid(n)[..., 0]
    Pr = P_o[..., 0]
    assert np.abs((Pd - Pd.mean()) - (Pr - Pr.mean())).max() <= 1e-6 * max(np.abs(Pr - Pr.mean()).max(), 1e-3)


def test_taylor_vortex_second_order(gpu):
    """Exact 2-D Taylor vortex extruded in z (reference Tutorials/TaylorGreen/benchmarks/EXACT_3D.F:75-119,
    prob.c = 0): L2 velocity error at fixed time converges with order ~2 (32^3 -> 64^3)."""
    lib = gpu
    from iamr_amd import ns as N
    visc = 1e-2
    T = 0.1
    errs = []
    for N_ in (32, 64):
        n = (N_,) * 3
        g = lib.Geom.make(n)
        lay = lib.Layout.single(n)
        ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.5, visc_coef=visc, init_iter=2, fixed_dt=T / (N_ // 4)))
        ns.init_taylorgreen(1.0, 1.0, 1.0, 0.0, 1.0)
        ns.post_init(T)
        for _ in range(N_ // 4):
            ns.step()
        assert abs(ns.time - T) < 1e-12
        S = ns.data(N.NavierStokes.S_NEW).gather_valid(n)
        x = (np.arange(N_) + 0.5) / N_
        X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
        dec = np.exp(-8 * np.pi ** 2 * visc * T)
        ue = np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) * dec
        ve = -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y) * dec
        errs.append(np.sqrt(((S[..., 0] - ue) ** 2 + (S[..., 1] - ve) ** 2).mean()))
        assert np.abs(S[..., 2]).max() < 1e-10      # w stays zero (z-uniform flow)
    order = np.log2(errs[0] / errs[1])
    assert 1.7 <= order <= 2.6, (errs, order)
