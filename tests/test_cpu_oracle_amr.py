"""Pins of the multi-level oracle (oracle/orc_amr.c) and of the round-2 additions to the single-level one that do not need a GPU:
known answers and invariants that do not depend on the product (usable pins while AMReX / AMReX-Hydro are absent)."""
import ctypes as C
import numpy as np

import orc


def _amr(n0, fine, state_fn, **kw):
    a = orc.OrcAmr(orc.geom([n0] * 3), orc.ns_params(**kw), orc.mg_opts(), [[], fine])
    for l in range(2):
        a.set_state(l, state_fn(*a.cell_centres(l)))
    return a


def _uniform(X, Y, Z):
    S = np.zeros(X.shape + (5,), order="F")
    S[..., 0], S[..., 1], S[..., 2], S[..., 3] = 1.0, 0.5, 0.25, 1.0
    S[..., 4] = np.exp(-((X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.5) ** 2) / 0.02)
    return S


def _composite(a, comp):
    S0, S1, c1 = a.state(0), a.state(1), a.cov(1)
    return (S0[..., comp] * ~c1[::2, ::2, ::2]).sum() * np.prod(a.dx(0)) + (S1[..., comp] * c1).sum() * np.prod(a.dx(1))


def test_uniform_flow_is_preserved_and_tracer_is_conserved_across_the_interface():
    """A uniform velocity field with constant density is an exact solution of every piece of the algorithm: predictor, both MAC
    solves (coarse/fine Dirichlet data), create_umac_grown, level projections, reflux, MAC sync and sync projection must leave it
    alone to round-off, while the tracer blob is advected through the coarse/fine interface without losing mass."""
    a = _amr(8, [([4, 4, 4], [11, 11, 11])], _uniform, cfl=0.7, init_iter=2)
    a.post_init()
    m0 = _composite(a, 4)
    for _ in range(3):
        a.step()
    for l in range(2):
        S, c = a.state(l), a.cov(l)
        assert abs(S[..., 0] - 1.0)[c].max() < 1e-13 and abs(S[..., 1] - 0.5)[c].max() < 1e-13 and abs(S[..., 2] - 0.25)[c].max() < 1e-13
        assert abs(S[..., 3] - 1.0)[c].max() < 1e-13
        assert abs(a.fab(l, 2).a).max() < 1e-12          # no pressure develops
    assert abs(_composite(a, 4) - m0) < 1e-14
    assert abs(_composite(a, 3) - 1.0) < 1e-14


def test_composite_projection_is_second_order_consistent():
    """The cell-centred Taylor-Green field is discretely divergence free on each level separately; the composite divergence at the
    coarse/fine nodes is not (different quadrature of the two sides), so the initial composite projection changes the field -- by
    O(h^2) if the composite operator and right-hand side are consistent."""
    ch = []
    for n0 in (8, 16):
        lo, hi = n0 // 4, n0 // 4 + n0 - 1
        a = _amr(n0, [([lo] * 3, [hi] * 3)], lambda X, Y, Z: orc.taylorgreen_state(X, Y, Z, c=0.0), init_iter=0)
        a.post_init()
        e = 0.0
        for l in range(2):
            ex = orc.taylorgreen_state(*a.cell_centres(l), c=0.0)
            e = max(e, abs(a.state(l)[..., :3] - ex[..., :3])[a.cov(l)].max())
        ch.append(e)
    assert ch[0] < 0.05 and ch[1] < ch[0] / 3.0


def test_two_level_taylor_vortex_conserves_and_converges_to_tolerance():
    """two coarse steps of the exact Taylor vortex (prob.c = 0) on 16^3 + a refined patch: mass and tracer conserved to round-off,
    momentum to the tolerance of the sync solves, the sync solves converge, the error stays at the level of the single-level scheme"""
    a = _amr(16, [([4, 4, 4], [19, 19, 19])], lambda X, Y, Z: orc.taylorgreen_state(X, Y, Z, c=0.0), cfl=0.7, init_iter=2)
    a.post_init()
    m0 = [_composite(a, c) for c in range(5)]
    for _ in range(2):
        a.step()
        st = a.sync_stats()
        assert st.converged == 1 and st.resnorm <= 1e-10 * max(st.rhsnorm0, st.resnorm0) * 1.0001
    m1 = [_composite(a, c) for c in range(5)]
    assert abs(m1[3] - m0[3]) < 1e-13 and abs(m1[4] - m0[4]) < 1e-13
    for c in range(3):
        assert abs(m1[c] - m0[c]) < 1e-9
    for l in range(2):
        ex = orc.taylorgreen_state(*a.cell_centres(l), c=0.0)
        assert abs(a.state(l)[..., :3] - ex[..., :3])[a.cov(l)].max() < 0.03


def test_hydrostatic_initial_pressure():
    """NavierStokesBase::post_init_state (NavierStokesBase.cpp:2416-2426): with gravity the initial pressure projection makes
    grad p = rho g; a stably stratified fluid at rest then stays at rest (known answer, independent of the product)."""
    n = 16
    g = orc.geom([n] * 3, periodic=(1, 1, 0))
    p = orc.ns_params(cfl=0.5, gravity=-9.8, init_iter=2, phys_lo=[0, 0, 4], phys_hi=[0, 0, 4], init_dt=0.01)
    L = orc.lib()
    ns = C.c_void_p(L.orc_ns_create(C.byref(g), C.byref(p), C.byref(orc.mg_opts())))
    S = orc.from_cfab(L.orc_ns_fab(ns, 0))
    z = (np.arange(n) + 0.5) / n
    S.valid([n] * 3)[...] = 0.0
    S.valid([n] * 3)[..., 3] = (2.0 - z)[None, None, :]
    L.orc_ns_post_init(ns, C.c_double(-1.0))
    Gp = orc.from_cfab(L.orc_ns_fab(ns, 4)).valid([n] * 3)
    rho = orc.from_cfab(L.orc_ns_fab(ns, 0)).valid([n] * 3)[..., 3]
    assert abs(Gp[..., 2] - rho * (-9.8)).max() < 1e-11 and abs(Gp[..., :2]).max() < 1e-11
    for _ in range(2):
        L.orc_ns_step(ns)
    assert abs(orc.from_cfab(L.orc_ns_fab(ns, 0)).valid([n] * 3)[..., :3]).max() < 1e-10
    L.orc_ns_destroy(ns)


def test_stop_time_is_hit_exactly():
    """computeNewDt (NavierStokesBase.cpp:1008-1015): the last step is shortened to land on stop_time"""
    n = 8
    g = orc.geom([n] * 3)
    L = orc.lib()
    ns = C.c_void_p(L.orc_ns_create(C.byref(g), C.byref(orc.ns_params(cfl=0.7, init_iter=1)), C.byref(orc.mg_opts())))
    L.orc_ns_init_taylorgreen(ns, C.c_double(1.0), C.c_double(1.0), C.c_double(1.0), C.c_double(0.0), C.c_double(1.0))
    stop = 0.21
    L.orc_ns_post_init(ns, C.c_double(stop))
    for _ in range(50):
        if L.orc_ns_time(ns) >= stop - 1e-14:
            break
        L.orc_ns_step(ns)
    assert L.orc_ns_time(ns) == stop
    L.orc_ns_destroy(ns)


def test_three_level_uniform_flow_and_conservation():
    """three levels (sync operations on a refined level: mac_sync_solve with homogeneous coarse/fine data, CompAdd, SyncInterp with
    the accumulated ratio 4, SyncProjInterp): the uniform flow stays uniform to round-off and the tracer blob crossing both
    interfaces keeps its composite mass"""
    a = orc.OrcAmr(orc.geom([8] * 3), orc.ns_params(cfl=0.7, init_iter=2), orc.mg_opts(),
                   [[], [([2, 2, 2], [13, 13, 13])], [([10, 10, 10], [21, 21, 21])]])
    for l in range(3):
        a.set_state(l, _uniform(*a.cell_centres(l)))

    def comp(c):
        tot = 0.0
        for l in range(3):
            m = a.cov(l).copy()
            if l < 2:
                m &= ~a.cov(l + 1)[::2, ::2, ::2]
            tot += (a.state(l)[..., c] * m).sum() * np.prod(a.dx(l))
        return tot
    a.post_init()
    m0 = comp(4)
    for _ in range(2):
        a.step()
    for l in range(3):
        S, c = a.state(l), a.cov(l)
        for q, v in enumerate((1.0, 0.5, 0.25, 1.0)):
            assert abs(S[..., q] - v)[c].max() < 1e-13
        assert abs(a.fab(l, 2).a).max() < 1e-12
    assert abs(comp(4) - m0) < 1e-14 and abs(comp(3) - 1.0) < 1e-14


def test_ppm_is_exact_for_parabolic_data():
    """Known answer for the restated Colella-Woodward reconstruction (oracle/orc_godunov.c, Godunov_PPM): away from extrema and from the
    periodic wrap the limiters are inactive, the reconstructed parabola IS the data and the state traced to a face is the exact average
    of that parabola over the domain of dependence [x_f - u dt, x_f].  PLM (second order) does not reproduce it."""
    n = (32, 4, 4)
    g = orc.geom(n)
    L = orc.lib()
    h, dt = 1.0 / n[0], 0.4 / n[0]
    x = (np.arange(-3, n[0] + 3) + 0.5) * h
    a, b, c = 0.6, 0.3, 0.5                              # u(x) = a + b x + c x^2 > 0 and increasing on [0, 1]
    vel = orc.Fab(n, orc.CELL, 3, 3)
    vel.a[...] = 0.0
    vel.a[..., 0] = (a + b * x + c * x * x)[:, None, None]
    force = orc.Fab(n, orc.CELL, 1, 3)
    force.a[...] = 0.0
    res = {}
    for ppm in (1, 0):
        L.orc_godunov_set_ppm(ppm)
        um = [orc.Fab(n, orc.face(d), 1, 1) for d in range(3)]
        L.orc_extrap_vel_to_faces(C.byref(g), vel.ref(), force.ref(), orc.fabptrs(um), C.c_double(dt), orc.bcrecs(3), 0)
        res[ppm] = um[0].valid(n, orc.face(0))[:, 0, 0, 0].copy()
    L.orc_godunov_set_ppm(0)
    # the input values are point values q(x_i) = cell average of q - c h^2/12: PPM reconstructs q - c h^2/12.  Face i+1/2 takes the state
    # traced from cell i (u > 0): average over [x_f - s, x_f] with s = u_i dt
    i = np.arange(6, n[0] - 6)
    xf = (i + 1) * h
    ui = a + b * (i + 0.5) * h + c * ((i + 0.5) * h) ** 2
    s = ui * dt
    exact = a + b * (xf - s / 2) + c * (xf * xf - xf * s + s * s / 3) - c * h * h / 12
    assert np.abs(res[1][i + 1] - exact).max() < 1e-13
    assert np.abs(res[0][i + 1] - exact).max() > 1e-6


def test_cell_cons_interp_known_answers():
    """CellConservativeLinear(false) = IAMR's cell_cons_interp (coarse -> fine data of FillPatch, FillCoarsePatch and SyncInterp): exact for
    linear data, conservative (the fine cells of a coarse cell average to it) and bounded by the 27 neighbouring coarse values
    for any data -- per component, so that a rough component does not switch the slopes of a smooth one off."""
    nc, r = (8, 8, 8), 2
    nf = tuple(v * r for v in nc)
    L = orc.lib()
    gc = orc.geom(nc)
    rng = np.random.default_rng(5)
    xc = [(np.arange(-4, nc[d] + 4) + 0.5) / nc[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*xc, indexing="ij")
    cf = orc.Fab(nc, orc.CELL, 4, 2)
    cf.a[..., 0] = 1.0 + 2.0 * X - 3.0 * Y + 0.5 * Z                        # linear (not periodic: only interior cells are checked)
    cf.a[..., 1] = rng.standard_normal(X.shape)                             # rough
    L.orc_fill_periodic(cf.ref(), C.byref(gc), orc.i3(orc.CELL))
    cf.a[..., 0] = 1.0 + 2.0 * X - 3.0 * Y + 0.5 * Z
    ff = orc.Fab(nf, orc.CELL, 0, 2)
    ff.a[...] = -99.0
    empty_lo, empty_hi = (1, 1, 1), (0, 0, 0)
    L.orc_fill_coarse_fine(ff.ref(), orc.i3((0, 0, 0)), orc.i3(tuple(v - 1 for v in nf)), orc.i3(empty_lo), orc.i3(empty_hi), cf.ref(),
                           orc.i3((0, 0, 0)), orc.i3(tuple(v - 1 for v in nc)), orc.i3((1, 1, 1)), r, orc.bcrecs(2))
    F = ff.a
    xf = [(np.arange(nf[d]) + 0.5) / nf[d] for d in range(3)]
    Xf, Yf, Zf = np.meshgrid(*xf, indexing="ij")
    inner = (slice(2, -2),) * 3
    assert np.abs(F[..., 0] - (1.0 + 2.0 * Xf - 3.0 * Yf + 0.5 * Zf))[inner].max() < 1e-13
    Cv = cf.a[4:-4, 4:-4, 4:-4, :]
    avg = F.reshape(nc[0], r, nc[1], r, nc[2], r, 2).mean(axis=(1, 3, 5))
    assert np.abs(avg - Cv).max() < 1e-13                                    # conservative, both components
    # no new extrema: every fine value lies between the min and max of the 27 coarse neighbours of its coarse cell
    P = cf.a[3:-3, 3:-3, 3:-3, 1]
    lo = np.full(nc, np.inf); hi = np.full(nc, -np.inf)
    for dx in range(3):
        for dy in range(3):
            for dz in range(3):
                blk = P[dx:dx + nc[0], dy:dy + nc[1], dz:dz + nc[2]]
                lo, hi = np.minimum(lo, blk), np.maximum(hi, blk)
    Fr = F[..., 1].reshape(nc[0], r, nc[1], r, nc[2], r)
    assert (Fr >= lo[:, None, :, None, :, None] - 1e-13).all() and (Fr <= hi[:, None, :, None, :, None] + 1e-13).all()
    assert np.abs(Fr - Cv[:, None, :, None, :, None, 1]).max() > 1e-3          # the slopes of the rough component are not all limited to zero
