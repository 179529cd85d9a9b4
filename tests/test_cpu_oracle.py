"""CPU suite (-m "not gpu"): pins of the oracle (oracle/liborc.so).

The reference commits no golden vectors for this path and its arithmetic lives in un-vendored, unpinned
AMReX / AMReX-Hydro -> PARITY UNPINNED.  What CAN be pinned, and is pinned here:
  * exact Taylor vortex (reference Tutorials/TaylorGreen/benchmarks/EXACT_3D.F:75-119) + 2nd-order convergence
  * discrete eigen-answers of the 7-point and the Q1 27-point operators
  * invariants: MAC divergence, density conservation, uniform-state advection, solver residual reduction
  * the committed fixture tests/golden/taylorgreen16_oracle.npz (regression pin of the oracle itself)
"""
import ctypes as C
import os
import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def cc(n, ng):
    return [(np.arange(-ng, n[d] + ng) + 0.5) / n[d] for d in range(3)]


def test_abec_eigen_answer_and_solve(orc):
    L = orc.lib()
    N = 16
    n = (N,) * 3
    g = orc.geom(n)
    dx = 1.0 / N
    b = [orc.Fab(n, orc.face(d), 0, 1, fill=1.0) for d in range(3)]
    lev = orc.abec_level(g, b)
    X, Y, Z = np.meshgrid(*cc(n, 1), indexing="ij")
    phi = orc.Fab(n, orc.CELL, 1, 1)
    phi.a[..., 0] = np.sin(2 * np.pi * X) * np.sin(4 * np.pi * Y) * np.cos(2 * np.pi * Z)
    y = orc.Fab(n, orc.CELL, 0, 1)
    L.orc_abec_apply(C.byref(lev), y.ref(), phi.ref())
    lam = (4 / dx ** 2) * (np.sin(np.pi * dx) ** 2 + np.sin(2 * np.pi * dx) ** 2 + np.sin(np.pi * dx) ** 2)
    assert np.abs(y.a[..., 0] - lam * phi.valid(n)[..., 0]).max() <= 1e-11 * lam
    rhs = orc.Fab(n, orc.CELL, 0, 1)
    rhs.a[...] = lam * phi.valid(n)
    sol = orc.Fab(n, orc.CELL, 1, 1)
    st = orc.CMgStats()
    o = orc.mg_opts()
    z = orc.i3([0, 0, 0])
    L.orc_abec_solve(C.byref(lev), sol.ref(), rhs.ref(), z, z, C.c_double(1e-12), C.c_double(1e-16), C.byref(o), C.byref(st))
    assert st.converged == 1 and st.iters <= 12
    s = sol.valid(n)[..., 0]
    assert np.abs((s - s.mean()) - phi.valid(n)[..., 0]).max() <= 1e-12


def test_nodal_eigen_answer(orc):
    L = orc.lib()
    N = 16
    n = (N,) * 3
    g = orc.geom(n)
    h = 1.0 / N
    x = np.arange(-1, N + 2) / N
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
    k = (1, 2, 1)
    phi = orc.Fab(n, orc.NODE, 1, 1)
    phi.a[..., 0] = np.cos(2 * np.pi * (k[0] * X + k[1] * Y + k[2] * Z))
    sig = orc.Fab(n, orc.CELL, 1, 1, fill=1.0)
    y = orc.Fab(n, orc.NODE, 0, 1)
    L.orc_nodal_adotx(C.byref(g), y.ref(), phi.ref(), sig.ref())
    th = [2 * np.pi * kk * h for kk in k]
    S = [(2 - 2 * np.cos(t)) / h ** 2 for t in th]
    M = [(2 + np.cos(t)) / 3 for t in th]
    lam = -(S[0] * M[1] * M[2] + S[1] * M[0] * M[2] + S[2] * M[0] * M[1])
    assert np.abs(y.a[..., 0] - lam * phi.valid(n, orc.NODE)[..., 0]).max() <= 1e-11 * abs(lam)


def test_nodal_smoothers_agree_and_projection_is_second_order(orc):
    """colour GS (product ordering), lexicographic GS (reference CPU ordering) and Jacobi converge to the same phi;
    the nodal divergence left by the approximate projection is O(h^2)."""
    L = orc.lib()
    z = orc.i3([0, 0, 0])
    left = []
    for N in (8, 16):
        n = (N,) * 3
        g = orc.geom(n)
        X, Y, Z = np.meshgrid(*cc(n, 1), indexing="ij")
        sig = orc.Fab(n, orc.CELL, 1, 1)
        sig.a[..., 0] = 1.0 / (1.0 + 0.5 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y))
        vel = orc.Fab(n, orc.CELL, 1, 3)
        vel.a[..., 0] = np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) + 0.3 * np.cos(2 * np.pi * Z)
        vel.a[..., 1] = -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y) + 0.2 * np.sin(2 * np.pi * Y)
        vel.a[..., 2] = 0.5 * np.sin(2 * np.pi * Z) * np.cos(2 * np.pi * X)
        ref = None
        for sm in (0, 1, 2):
            v = vel.copy()
            p = orc.Fab(n, orc.NODE, 1, 1)
            st = orc.CMgStats()
            o = orc.mg_opts(nodal_smoother=sm)
            L.orc_nodal_project(C.byref(g), v.ref(), p.ref(), sig.ref(), z, z, C.c_double(1e-12), C.c_double(1e-16), C.byref(o), C.byref(st))
            assert st.converged == 1
            pv = p.valid(n, orc.NODE)[..., 0]
            pv = pv - pv.mean()
            if ref is None:
                ref = pv
                L.orc_fill_periodic(v.ref(), C.byref(g), orc.i3(orc.CELL))
                r = orc.Fab(n, orc.NODE, 0, 1)
                L.orc_nodal_divu(C.byref(g), r.ref(), v.ref())
                left.append(np.abs(r.a).max())
            else:
                assert np.abs(pv - ref).max() <= 1e-10
    assert left[1] < 0.5 * left[0]


def test_mac_projection_invariant(orc):
    L = orc.lib()
    N = 16
    n = (N,) * 3
    g = orc.geom(n)
    rng = np.random.default_rng(1)
    um = []
    for d in range(3):
        t = orc.face(d)
        f = orc.Fab(n, t, 1, 1)
        f.a[...] = rng.standard_normal(f.a.shape)
        hi = [slice(None)] * 3
        lo = [slice(None)] * 3
        hi[d] = 1 + N
        lo[d] = 1
        f.a[tuple(hi)] = f.a[tuple(lo)]
        L.orc_fill_periodic(f.ref(), C.byref(g), orc.i3(t))
        um.append(f)
    X, Y, Z = np.meshgrid(*cc(n, 1), indexing="ij")
    rho = orc.Fab(n, orc.CELL, 1, 1)
    rho.a[..., 0] = 1.0 + 0.4 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Z)
    phi = orc.Fab(n, orc.CELL, 1, 1)
    st = orc.CMgStats()
    o = orc.mg_opts(maxorder=4)
    z = orc.i3([0, 0, 0])
    L.orc_mac_project(C.byref(g), orc.fabptrs(um), rho.ref(), None, phi.ref(), C.c_double(200.0), z, z, C.c_double(1e-12),
                      C.c_double(1e-16), C.byref(o), C.byref(st))
    div = orc.Fab(n, orc.CELL, 0, 1)
    L.orc_mac_divergence(C.byref(g), div.ref(), orc.fabptrs(um))
    assert st.converged == 1
    assert np.abs(div.a).max() <= 1e-11 * st.rhsnorm0 * 10


def test_godunov_uniform_state_and_symmetry(orc):
    L = orc.lib()
    N = 12
    n = (N,) * 3
    g = orc.geom(n)
    # uniform state advected by a non-trivial divergence-free velocity: aofs(conservative) = q div(u) = 0, convective = 0
    X, Y, Z = np.meshgrid(*cc(n, 3), indexing="ij")
    S = orc.Fab(n, orc.CELL, 3, 2, fill=1.7)
    um = []
    for d in range(3):
        f = orc.Fab(n, orc.face(d), 1, 1, fill=[0.3, -0.2, 0.5][d])
        um.append(f)
    aofs = orc.Fab(n, orc.CELL, 0, 2)
    ic = (C.c_int * 2)(1, 0)
    L.orc_compute_aofs(C.byref(g), aofs.ref(), 0, S.ref(), 2, None, None, orc.fabptrs(um), ic, C.c_double(0.02), orc.bcrecs(2), 0, 0, None, None)
    assert np.abs(aofs.a).max() <= 1e-12
    # direction-permutation symmetry of the generic-direction code
    def run(perm):
        U = np.array([1.0, 0.5, 0.25])[list(perm)]
        kk = np.array([1, 2, 1])[list(perm)]
        ph = 2 * np.pi * (kk[0] * X + kk[1] * Y + kk[2] * Z)
        Sf = orc.Fab(n, orc.CELL, 3, 1)
        Sf.a[..., 0] = np.sin(ph)
        ums = [orc.Fab(n, orc.face(d), 1, 1, fill=U[d]) for d in range(3)]
        a = orc.Fab(n, orc.CELL, 0, 1)
        L.orc_compute_aofs(C.byref(g), a.ref(), 0, Sf.ref(), 1, None, None, orc.fabptrs(ums), (C.c_int * 1)(0), C.c_double(0.4 / N),
                           orc.bcrecs(1), 0, 0, None, None)
        return a.a.copy()
    a = run((0, 1, 2))
    b = run((2, 0, 1))
    assert np.abs(np.transpose(a, (2, 0, 1, 3)) - b).max() <= 1e-13


def run_tg(orc, N, T, visc=1e-2, c=0.0, nsteps=None):
    L = orc.lib()
    n = (N,) * 3
    g = orc.geom(n)
    p = orc.CNsParams()
    L.orc_ns_default_params(C.byref(p))
    p.cfl = 0.5
    p.visc_coef = visc
    if nsteps is None:
        nsteps = N // 4
        p.fixed_dt = T / nsteps
    o = orc.mg_opts()
    s = C.c_void_p(L.orc_ns_create(C.byref(g), C.byref(p), C.byref(o)))
    L.orc_ns_init_taylorgreen(s, C.c_double(1.0), C.c_double(1.0), C.c_double(1.0), C.c_double(c), C.c_double(1.0))
    L.orc_ns_post_init(s, C.c_double(T if T else -1.0))
    for _ in range(nsteps):
        L.orc_ns_step(s)
    S = orc.from_cfab(L.orc_ns_fab(s, 0)).valid(n).copy()
    P = orc.from_cfab(L.orc_ns_fab(s, 2)).valid(n, orc.NODE).copy()
    t = L.orc_ns_time(s)
    L.orc_ns_destroy(s)
    return S, P, t


def test_taylor_vortex_exact_solution_second_order(orc):
    """exact solution u = sin(2 pi x) cos(2 pi y) exp(-8 pi^2 nu t), v = -cos sin exp(...) (EXACT_3D.F:75-119, prob.c = 0)"""
    visc, T = 1e-2, 0.1
    errs = []
    for N in (8, 16):
        S, P, t = run_tg(orc, N, T, visc)
        assert abs(t - T) < 1e-12
        x = (np.arange(N) + 0.5) / N
        X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
        dec = np.exp(-8 * np.pi ** 2 * visc * T)
        ue = np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) * dec
        ve = -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y) * dec
        errs.append(np.sqrt(((S[..., 0] - ue) ** 2 + (S[..., 1] - ve) ** 2).mean()))
        assert np.abs(S[..., 2]).max() < 1e-12
        assert abs(S[..., 3].sum() - S[..., 3].size) < 1e-9          # mass conserved to round-off
    order = np.log2(errs[0] / errs[1])
    assert 1.6 <= order <= 2.8, (errs, order)


def test_golden_fixture_regression(orc):
    """the oracle reproduces its committed fixture (generated by tests/golden/make_golden.py)"""
    path = os.path.join(HERE, "golden", "taylorgreen16_oracle.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    gold = np.load(path)
    S, P, t = run_tg(orc, 16, None, visc=float(gold["visc"]), c=float(gold["c"]), nsteps=int(gold["nsteps"]))
    assert abs(t - float(gold["time"])) <= 1e-14
    assert np.abs(S - gold["S"]).max() <= 1e-12


def run_ldc(orc, n, phys_lo, phys_hi, nsteps, extrap_scale=1.0, init=None, per=(0, 0, 0), **kw):
    """LidDrivenCavity set-up of Exec/run3d/regtest.3d.lid_driven_cavity:5-46 (lid zhi.velocity = 1 0 0)"""
    L = orc.lib()
    g = orc.geom(n, periodic=per)
    p = orc.CNsParams()
    L.orc_ns_default_params(C.byref(p))
    par = dict(cfl=0.3, visc_coef=0.01, init_dt=0.0140625, init_shrink=0.3, init_iter=3, tracer_diff_coef=0.001)
    par.update(kw)
    for k, v in par.items():
        setattr(p, k, v)
    for d in range(3):
        p.phys_lo[d] = 0 if per[d] else phys_lo[d]
        p.phys_hi[d] = 0 if per[d] else phys_hi[d]
    p.wall_vel_hi[2 * 3 + 0] = 1.0
    o = orc.mg_opts()
    L.orc_ns_test_set_extrap_scale(C.c_double(extrap_scale))
    s = C.c_void_p(L.orc_ns_create(C.byref(g), C.byref(p), C.byref(o)))
    L.orc_ns_init_rest(s, C.c_double(1.0))
    if init is not None:
        orc.from_cfab(L.orc_ns_fab(s, 0)).a[1:-1, 1:-1, 1:-1, :] = init
    L.orc_ns_post_init(s, C.c_double(-1.0))
    dts = [L.orc_ns_step(s) for _ in range(nsteps)]
    S = orc.from_cfab(L.orc_ns_fab(s, 0)).valid(n).copy()
    t = L.orc_ns_time(s)
    L.orc_ns_destroy(s)
    L.orc_ns_test_set_extrap_scale(C.c_double(1.0))
    return S, t, dts


def test_lid_driven_cavity_slip_side_walls_stay_two_dimensional(orc):
    """cavity with slip walls in y on both sides: the flow driven by the x-moving lid has v = 0 and no y dependence;
    first dt = init_shrink * init_dt, later dts grow by change_max until the CFL limit binds"""
    n = (12, 8, 12)
    S, t, dts = run_ldc(orc, n, (5, 4, 5), (5, 4, 5), 4)
    assert abs(dts[0] - 0.3 * 0.0140625) < 1e-15
    assert all(abs(dts[i + 1] / dts[i] - 1.1) < 1e-12 for i in range(3))
    assert np.abs(S[..., 0]).max() > 0.03
    assert np.abs(S[..., 1]).max() < 1e-11
    assert np.abs(S - S[:, :1]).max() < 1e-11
    assert np.abs(S[..., 3] - 1.0).max() < 1e-10 and np.abs(S[..., 4]).max() == 0.0


def test_wall_ghost_forcing_is_immaterial(orc):
    """the viscous-term ghost cells outside the walls (Extrapolater::FirstOrderExtrap in the reference; restated as a
    nearest-cell copy) only feed Godunov states that the wall BC overrides: scaling them by 1e3 changes nothing"""
    n = (8, 8, 8)
    rng = np.random.default_rng(11)
    x = [(np.arange(n[d]) + 0.5) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*x, indexing="ij")
    init = np.zeros(n + (5,))
    init[..., 0] = np.sin(np.pi * X) * np.sin(np.pi * Y) * np.sin(np.pi * Z)
    init[..., 1] = 0.4 * np.sin(2 * np.pi * X) * np.sin(np.pi * Y) * np.sin(np.pi * Z)
    init[..., 2] = 0.2 * rng.random(n) * np.sin(np.pi * X) * np.sin(np.pi * Y) * np.sin(np.pi * Z)
    init[..., 3] = 1.0
    init[..., 4] = np.exp(-30.0 * ((X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.5) ** 2))
    kw = dict(cfl=0.5, visc_coef=0.05, tracer_diff_coef=0.02, init_iter=1)
    a, _, _ = run_ldc(orc, n, (4, 4, 5), (5, 5, 5), 2, 1.0, init, **kw)
    b, _, _ = run_ldc(orc, n, (4, 4, 5), (5, 5, 5), 2, 1.0e3, init, **kw)
    assert np.abs(a[..., 0]).max() > 0.1
    assert np.array_equal(a, b)


def test_momentum_form_and_conservative_tracer(orc):
    """ns.do_mom_diff / ns.do_cons_trac (NS_setup.cpp:297-310) on a periodic box: (a) with uniform density the momentum form differs from
    the convective form only in the truncation error of the transverse terms ((T_avg - q) du/dx instead of 0), which shrinks under
    refinement; (b) with variable density it is a different scheme; (c) a conservative tracer S = c rho stays c rho (same conservative
    update, limiters are homogeneous) and sum(S) is conserved, with and without diffusion of S/rho"""
    per = (1, 1, 1)
    kw = dict(cfl=0.5, visc_coef=0.02, tracer_diff_coef=0.0, init_iter=1, init_dt=-1.0, init_shrink=1.0, fixed_dt=0.01)
    diffs = []
    for N in (8, 16):
        n = (N, N, N)
        x = [(np.arange(n[d]) + 0.5) / n[d] for d in range(3)]
        X, Y, Z = np.meshgrid(*x, indexing="ij")
        init = np.zeros(n + (5,))
        init[..., 0] = np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) + 0.2 * np.sin(2 * np.pi * Z)
        init[..., 1] = -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y)
        init[..., 2] = 0.3 * np.sin(2 * np.pi * X) * np.sin(2 * np.pi * Y)
        init[..., 3] = 1.0
        init[..., 4] = np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Z)
        a, _, _ = run_ldc(orc, n, (0, 0, 0), (0, 0, 0), 2, 1.0, init, per, **kw)
        b, _, _ = run_ldc(orc, n, (0, 0, 0), (0, 0, 0), 2, 1.0, init, per, do_mom_diff=1, **kw)
        assert np.abs(a[..., 0]).max() > 0.5
        diffs.append(np.abs(a - b).max())
    assert 0 < diffs[1] < 0.4 * diffs[0] < 2e-3
    del kw["fixed_dt"]
    n = (8, 8, 8)
    x = [(np.arange(n[d]) + 0.5) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*x, indexing="ij")
    init = init[::2, ::2, ::2].copy()
    init[..., 3] = 1.0 + 0.5 * np.exp(-20.0 * ((X - 0.5) ** 2 + (Y - 0.4) ** 2 + (Z - 0.6) ** 2))
    a, _, _ = run_ldc(orc, n, (0, 0, 0), (0, 0, 0), 2, 1.0, init, per, **kw)
    b, _, _ = run_ldc(orc, n, (0, 0, 0), (0, 0, 0), 2, 1.0, init, per, do_mom_diff=1, **kw)
    assert np.abs(a[..., :3] - b[..., :3]).max() > 1e-3
    init[..., 4] = 0.75 * init[..., 3]
    for diff in (0.0, 0.05):
        kw["tracer_diff_coef"] = diff
        c, _, _ = run_ldc(orc, n, (0, 0, 0), (0, 0, 0), 3, 1.0, init, per, do_mom_diff=1, do_cons_trac=1, **kw)
        assert np.abs(c[..., 4] - 0.75 * c[..., 3]).max() < 1e-12
        assert np.abs(c[..., 3] - init[..., 3]).max() > 1e-3
        assert abs(c[..., 4].sum() - init[..., 4].sum()) < 1e-10 * init[..., 4].sum()
    init[..., 4] = init[..., 3] * (0.5 + 0.3 * np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Z))
    c, _, _ = run_ldc(orc, n, (0, 0, 0), (0, 0, 0), 3, 1.0, init, per, do_cons_trac=1, **kw)       # diffusive, non-trivial concentration
    assert abs(c[..., 4].sum() - init[..., 4].sum()) < 1e-9 * init[..., 4].sum()
    q0, q1 = init[..., 4] / init[..., 3], c[..., 4] / c[..., 3]
    assert q1.max() < q0.max() and q1.min() > q0.min()                                          # diffusion of q = S/rho


def test_golden_fixture_regression_lid_driven_cavity(orc):
    path = os.path.join(HERE, "golden", "liddrivencavity16_oracle.npz")
    if not os.path.exists(path):
        pytest.skip("fixture not generated")
    gold = np.load(path)
    S, t, dts = run_ldc(orc, (16, 16, 16), (4, 4, 5), (5, 5, 5), int(gold["nsteps"]))
    assert abs(t - float(gold["time"])) <= 1e-14
    assert np.abs(S - gold["S"]).max() <= 1e-12


def test_nodal_solve_with_dirichlet_nodes(orc):
    """orc_nodal_solve_cov: (a) all-Dirichlet box recovers a manufactured solution; (b) a level covering a sub-box of a periodic
    domain changes only its interior nodes and drives the residual there to the tolerance"""
    import ctypes as C
    L = orc.lib()
    L.orc_nodal_solve_cov.restype = None
    n = (16, 16, 16)
    rng = np.random.default_rng(0)
    sig = orc.Fab(n, orc.CELL, 1, 1)
    sig.a[...] = 1.0 + 0.5 * rng.random(sig.a.shape)
    # (a)
    g = orc.geom(n, periodic=(0, 0, 0))
    D = (C.c_int * 3)(101, 101, 101)
    x = np.arange(0, n[0] + 1) / n[0]
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
    ex = np.sin(np.pi * X) * np.sin(np.pi * Y) * np.sin(np.pi * Z)
    phi_e = orc.Fab(n, orc.NODE, 1, 1)
    phi_e.a[1:-1, 1:-1, 1:-1, 0] = ex
    rhs = orc.Fab(n, orc.NODE, 0, 1)
    L.orc_nodal_adotx(C.byref(g), rhs.ref(), phi_e.ref(), sig.ref())
    phi = orc.Fab(n, orc.NODE, 1, 1)
    o = orc.mg_opts()
    st = orc.CMgStats()
    L.orc_nodal_solve_cov(C.byref(g), phi.ref(), rhs.ref(), sig.ref(), D, D, None, C.c_double(1e-11), C.c_double(0.0), C.byref(o), C.byref(st))
    assert st.converged and st.iters <= 6
    assert np.abs(phi.a[1:-1, 1:-1, 1:-1, 0] - ex).max() < 1e-10
    # (b)
    g2 = orc.geom(n)
    P = (C.c_int * 3)(0, 0, 0)
    cov = orc.Fab(n, orc.CELL, 0, 1)
    cov.a[4:12, 4:12, 2:14, 0] = 1.0
    phi = orc.Fab(n, orc.NODE, 1, 1)
    phi.a[...] = rng.random(phi.a.shape)
    phi0 = phi.a.copy()
    rhs = orc.Fab(n, orc.NODE, 0, 1)
    rhs.a[...] = rng.standard_normal(rhs.a.shape)
    o = orc.mg_opts(max_coarsening_level=1)
    L.orc_nodal_solve_cov(C.byref(g2), phi.ref(), rhs.ref(), sig.ref(), P, P, cov.ref(), C.c_double(1e-11), C.c_double(0.0), C.byref(o), C.byref(st))
    assert st.converged
    chg = np.abs(phi.a - phi0)[1:-1, 1:-1, 1:-1, 0] > 0
    interior = np.zeros_like(chg)
    interior[5:12, 5:12, 3:14] = True
    assert (chg & ~interior).sum() == 0 and chg.sum() == interior.sum()
    sg = sig.copy()
    sg.a[1:-1, 1:-1, 1:-1, 0] *= cov.a[..., 0]
    y = orc.Fab(n, orc.NODE, 0, 1)
    L.orc_nodal_adotx(C.byref(g2), y.ref(), phi.ref(), sg.ref())
    r = (rhs.a - y.a)[..., 0]
    assert np.abs(r[interior]).max() <= 1e-11 * st.resnorm0 * 10


def test_inflow_outflow_channel_conserves_the_inflow_flux(orc):
    """oracle with Inflow (1) / Outflow (2) in x, no-slip y walls, periodic z: the volume flux through every cross-section
    stays the inflow flux (MAC and nodal projections with Neumann/inflow and Dirichlet faces), the tracer enters with its
    inflow value, the density stays 1"""
    L = orc.lib()
    n = (32, 16, 4)
    g = orc.geom(n, probhi=(2.0, 1.0, 0.25), periodic=(0, 0, 1))
    p = orc.CNsParams()
    L.orc_ns_default_params(C.byref(p))
    for k, v in dict(cfl=0.5, visc_coef=0.05, init_shrink=0.3, init_iter=2).items():
        setattr(p, k, v)
    p.phys_lo[0], p.phys_hi[0], p.phys_lo[1], p.phys_hi[1] = 1, 2, 5, 5
    p.wall_vel_lo[0] = 1.0
    p.scal_bc_lo[0], p.scal_bc_lo[1] = 1.0, 0.5
    o = orc.mg_opts()
    s = C.c_void_p(L.orc_ns_create(C.byref(g), C.byref(p), C.byref(o)))
    assert s.value
    L.orc_ns_init_rest(s, C.c_double(1.0))
    orc.from_cfab(L.orc_ns_fab(s, 0)).a[1:-1, 1:-1, 1:-1, 0] = 1.0
    L.orc_ns_post_init(s, C.c_double(-1.0))
    for _ in range(6):
        L.orc_ns_step(s)
    S = orc.from_cfab(L.orc_ns_fab(s, 0)).valid(n).copy()
    L.orc_ns_destroy(s)
    assert np.abs(S[..., 0].mean(axis=(1, 2)) - 1.0).max() <= 1e-9
    assert np.abs(S[..., 3] - 1.0).max() <= 1e-12
    assert 0.05 < S[..., 4].max() <= 0.5 + 1e-12
    assert S[2, 0, 0, 0] < 0.8 < 1.1 < S[8, 8, 0, 0]          # the no-slip walls retard, the core accelerates
    # outflow on a side face with gravity: the outflow nodes hold the hydrostatic pressure of the column next to the face
    # (Projection::set_outflow_bcs / computeRhoG, Projection.cpp:1721-2370); uniform density: p = -g rho (z_top - z) exactly
    n = (16, 8, 8)
    g = orc.geom(n, probhi=(2.0, 1.0, 1.0), periodic=(0, 0, 0))
    p.gravity = -2.0
    p.phys_lo[2], p.phys_hi[2] = 4, 4
    s = C.c_void_p(L.orc_ns_create(C.byref(g), C.byref(p), C.byref(o)))
    assert s.value
    L.orc_ns_init_rest(s, C.c_double(1.0))
    orc.from_cfab(L.orc_ns_fab(s, 0)).a[1:-1, 1:-1, 1:-1, 0] = 1.0
    L.orc_ns_post_init(s, C.c_double(-1.0))
    L.orc_ns_step(s)
    Pn = orc.from_cfab(L.orc_ns_fab(s, 2)).valid(n, orc.NODE)[..., 0].copy()
    S = orc.from_cfab(L.orc_ns_fab(s, 0)).valid(n).copy()
    L.orc_ns_destroy(s)
    z = np.arange(n[2] + 1) / n[2]
    assert np.abs(Pn[-1] - 2.0 * (1.0 - z)[None, :]).max() <= 1e-13
    assert np.abs(S[..., 2]).max() <= 1e-8                      # the hydrostatic state stays at rest vertically
    assert np.abs(S[..., 0].mean(axis=(1, 2)) - 1.0).max() <= 1e-9


def test_cf_abec_solve_is_second_order_on_a_refined_patch(orc):
    """orc_cf_interp_bndry + orc_abec_solve_cf: -lap(phi) = f on two refined boxes inside a periodic domain, coarse data = the
    exact solution at the coarse cell centres.  The multigrid converges in a few cycles and the error drops ~4x per refinement."""
    L = orc.lib()
    L.orc_abec_solve_cf.restype = None
    L.orc_cf_interp_bndry.restype = None

    def exact(nn):
        x = (np.arange(-1, nn + 1) + 0.5) / nn
        X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
        return np.sin(2 * np.pi * X) * np.sin(2 * np.pi * Y) * np.sin(2 * np.pi * Z) + 0.3 * np.cos(2 * np.pi * X)

    errs = []
    for nf in (16, 32):
        n, nc = (nf,) * 3, (nf // 2,) * 3
        g = orc.geom(n)
        q = nf // 4
        boxes = [((q, q, q), (3 * q - 1, 3 * q - 1, 2 * q - 1)), ((q, q, 2 * q), (3 * q - 1, 3 * q - 1, 3 * q - 1))]
        b = [orc.Fab(n, orc.face(d), 0, 1, fill=1.0) for d in range(3)]
        Lv = orc.abec_level(g, b, boxes=boxes)
        cphi = orc.Fab(nc, orc.CELL, 1, 1)
        cphi.a[..., 0] = exact(nc[0])
        x = (np.arange(nf) + 0.5) / nf
        X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
        rhs = orc.Fab(n, orc.CELL, 0, 1)
        rhs.a[..., 0] = 12 * np.pi ** 2 * np.sin(2 * np.pi * X) * np.sin(2 * np.pi * Y) * np.sin(2 * np.pi * Z) + 0.3 * 4 * np.pi ** 2 * np.cos(2 * np.pi * X)
        bcv = orc.Fab(n, orc.CELL, 1, 3)
        L.orc_cf_interp_bndry(C.byref(Lv), 2, cphi.ref(), bcv.ref())
        phi = orc.Fab(n, orc.CELL, 1, 1)
        o = orc.mg_opts(maxorder=3)
        st = orc.CMgStats()
        P = (C.c_int * 3)(0, 0, 0)
        L.orc_abec_solve_cf(C.byref(Lv), phi.ref(), rhs.ref(), P, P, bcv.ref(), C.c_double(1e-10), C.c_double(0.0), C.byref(o), C.byref(st))
        assert st.converged and st.iters <= 12
        cov = np.zeros(n, bool)
        for lo, hi in boxes:
            cov[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1] = True
        errs.append(np.abs(phi.a[1:-1, 1:-1, 1:-1, 0] - exact(nf)[1:-1, 1:-1, 1:-1])[cov].max())
        assert np.abs(phi.a[1:-1, 1:-1, 1:-1, 0][~cov]).max() == 0.0        # cells outside the level are never touched
    assert errs[1] < 0.3 * errs[0] and errs[1] < 3e-3, errs


def test_temperature_run_satisfies_the_divergence_constraint(orc):
    """ns.do_temp (regtest.3d.hotspot): temperature as the last state component, div U = S = div(lambda grad T) / (rho T)
    (NavierStokes::calc_divu, NavierStokes.cpp:1876-1958) carried with dS/dt as two more components of the state arrays.
    Pins of the restatement: (a) lambda = 0 gives S = 0 and, bit for bit, the velocity / density / tracer of a run without
    temperature; (b) the MAC velocity of a step satisfies div(u_mac) = S^n + dt/2 (dS/dt)^n in every cell (create_mac_rhs,
    NavierStokesBase.cpp:1038-1065) to the tolerance of the MAC solve; (c) dS/dt is the difference quotient of S."""
    L = orc.lib()
    n = (16, 16, 16)
    g = orc.geom(n, probhi=(1.0, 1.0, 1.0), periodic=(1, 1, 0))
    x = (np.arange(16) + 0.5) / 16
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
    rho = 1.0 / (1.0 + 0.5 * np.exp(-((X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.4) ** 2) / 0.02))

    def run(cond, do_temp, steps=2):
        p = orc.CNsParams()
        L.orc_ns_default_params(C.byref(p))
        for k, v in dict(cfl=0.5, visc_coef=0.01, init_iter=2, init_shrink=0.3, do_temp=do_temp, temp_cond_coef=cond, gravity=-1.0).items():
            setattr(p, k, v)
        p.phys_lo[2], p.phys_hi[2] = 4, 2                       # slip wall below, outflow on top
        o = orc.mg_opts()
        s = C.c_void_p(L.orc_ns_create(C.byref(g), C.byref(p), C.byref(o)))
        assert s.value
        L.orc_ns_init_rest(s, C.c_double(1.0))
        a = orc.from_cfab(L.orc_ns_fab(s, 0)).a
        a[1:-1, 1:-1, 1:-1, 3] = rho
        if do_temp:
            assert a.shape[-1] == 8                             # u v w rho tracer temp + divu dsdt
            a[1:-1, 1:-1, 1:-1, 5] = 1.0 / rho
        L.orc_ns_post_init(s, C.c_double(-1.0))
        dts = [L.orc_ns_step(s) for _ in range(steps)]
        S = orc.from_cfab(L.orc_ns_fab(s, 0)).valid(n).copy()
        So = orc.from_cfab(L.orc_ns_fab(s, 1)).valid(n).copy()
        um = [orc.from_cfab(L.orc_ns_fab(s, 6 + d)).a.copy() for d in range(3)]
        L.orc_ns_destroy(s)
        return S, So, um, dts

    S0, _, _, _ = run(0.0, 0)
    S1, _, _, _ = run(0.0, 1)
    assert np.array_equal(S0[..., :5], S1[..., :5]) and np.abs(S1[..., 6:]).max() == 0.0
    S2, So, um, dts = run(1.0e-3, 1)
    assert np.isfinite(S2).all() and np.abs(S2[..., 6]).max() > 1e-3
    h = 1.0 / 16
    div = ((um[0][2:-1, 1:-1, 1:-1, 0] - um[0][1:-2, 1:-1, 1:-1, 0]) + (um[1][1:-1, 2:-1, 1:-1, 0] - um[1][1:-1, 1:-2, 1:-1, 0])
           + (um[2][1:-1, 1:-1, 2:-1, 0] - um[2][1:-1, 1:-1, 1:-2, 0])) / h
    rhs = So[..., 6] + 0.5 * dts[-1] * So[..., 7]
    assert np.abs(div - rhs).max() <= 1e-9 * max(1.0, np.abs(rhs).max() / 1e-3)
    assert np.abs(S2[..., 7] - (S2[..., 6] - So[..., 6]) / dts[-1]).max() <= 1e-12
