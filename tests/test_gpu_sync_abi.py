"""GPU: the operator-boundary entries added in round 6 (missing 6) on caller-owned arrays, through the C-ABI:
  iamrx_mac_sync_compute / iamrx_mac_sync_compute_edge   MacProj::mac_sync_compute (Source/MacProj.H:60-94, MacProj.cpp:488-786)
  iamrx_syncreg_comp_add                                  SyncRegister::CompAdd (Source/SyncRegister.H:45, SyncRegister.cpp:302-348)
  iamrx_initial_velocity_project / _sync_project          Projection::initialVelocityProject / initialSyncProject (Source/Projection.H:116-134)
The hierarchy's own time step calls the same functions (AmrNS::mac_sync / post_init / ml_sync_project) and is compared with the oracle in
tests/test_gpu_amr_step.py; here each entry is checked against its defining formula built from entries that have their own oracle tests
(edge states of iamrx_godunov_compute_aofs, iamrx_syncreg_fine_add, iamrx_nodal_projection, the hierarchy's post_init)."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _smooth(n, seed, shape_add=(0, 0, 0), ng=0, amp=1.0):
    rng = np.random.default_rng(seed)
    ax = [np.arange(-ng, n[d] + shape_add[d] + ng) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    f = np.zeros(X.shape)
    for _ in range(4):
        k = rng.integers(1, 3, size=3); ph = rng.random(3) * 2 * np.pi; a = rng.standard_normal()
        f += a * np.sin(2 * np.pi * k[0] * X + ph[0]) * np.cos(2 * np.pi * k[1] * Y + ph[1]) * np.sin(2 * np.pi * k[2] * Z + ph[2])
    return amp * f


def _mf(lib, lay, typ, G, ng):
    """MultiFab from a periodic global array with ng ghost layers (origin -ng)"""
    m = lib.MultiFab(lay, typ, G.shape[3], ng)
    m.set_from_global(G, (-ng,) * 3)
    return m


@pytest.mark.parametrize("mom", [0, 1])
@pytest.mark.parametrize("boxes", [None, 16])
def test_mac_sync_compute_matches_its_definition(gpu, boxes, mom):
    """Vsync / Ssync -= -div(edge * Ucorr * area) / vol with the edge states of ComputeAofs traced with u_mac under the forcing
    (gravity rho e_z + visc - grad p) [/ rho] (MacProj.cpp:598-640, NavierStokesBase.cpp:4681-4683, 4826-4832)"""
    lib = gpu
    n = (32, 32, 32)
    g = lib.Geom.make(n)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    nscal, dt, grav = 2, 0.004, -9.8
    vel = np.stack([_smooth(n, 10 + c, ng=3) for c in range(3)], axis=-1)
    scal = np.stack([1.0 + 0.2 * _smooth(n, 20, ng=3), 0.5 + 0.3 * _smooth(n, 21, ng=3)], axis=-1)
    Svel = vel * scal[..., :1] if mom else vel
    visc = np.stack([0.1 * _smooth(n, 30 + c, ng=1) for c in range(3)], axis=-1)
    gp = np.stack([_smooth(n, 40 + c, ng=1) for c in range(3)], axis=-1)
    tfs = np.stack([np.zeros_like(visc[..., 0]), 0.05 * _smooth(n, 50, ng=1)], axis=-1)
    um = [_smooth(n, 60 + d, shape_add=tuple(1 if e == d else 0 for e in range(3)), ng=1)[..., None] for d in range(3)]
    uc = [0.1 * _smooth(n, 70 + d, shape_add=tuple(1 if e == d else 0 for e in range(3)), ng=1)[..., None] for d in range(3)]
    for d in range(3):          # periodic duplicates of the face arrays agree by construction (functions of x / n)
        pass
    Sv_d, Ss_d = _mf(lib, lay, lib.CELL, Svel, 3), _mf(lib, lay, lib.CELL, scal, 3)
    visc_d, gp_d, tfs_d = _mf(lib, lay, lib.CELL, visc, 1), _mf(lib, lay, lib.CELL, gp, 1), _mf(lib, lay, lib.CELL, tfs, 1)
    um_d = [_mf(lib, lay, lib.face(d), um[d], 1) for d in range(3)]
    uc_d = [_mf(lib, lay, lib.face(d), uc[d], 1) for d in range(3)]
    V0, S0 = _smooth(n, 80)[..., None] * np.ones(3), _smooth(n, 81)[..., None] * np.ones(nscal)
    Vs = lib.MultiFab(lay, lib.CELL, 3, 1); Vs.setval(0.0); Vs.set_from_global(np.pad(V0, ((1, 1),) * 3 + ((0, 0),)), (-1,) * 3)
    Ss = lib.MultiFab(lay, lib.CELL, nscal, 1); Ss.setval(0.0); Ss.set_from_global(np.pad(S0, ((1, 1),) * 3 + ((0, 0),)), (-1,) * 3)
    fv = [lib.MultiFab(lay, lib.face(d), 3, 0) for d in range(3)]
    fs = [lib.MultiFab(lay, lib.face(d), nscal, 0) for d in range(3)]
    icons = [1, 0]
    lib.mac_sync_compute(g, uc_d, Vs, Ss, Sv_d, Ss_d, nscal, gp_d, um_d, icons, dt, visc_vel=visc_d, tforce_scal=tfs_d, do_mom_diff=mom, gravity=grav,
                         flux_vel=fv, flux_scal=fs)
    # ---- the definition, from the edge states of the (oracle-tested) advection entry
    rho1 = scal[2:-2, 2:-2, 2:-2, :1]
    gterm = np.zeros_like(visc)
    gterm[..., 2] = grav * rho1[..., 0]
    tfv = (gterm + visc) - gp                       # the order of the entry's expression
    if not mom:
        tfv = tfv / rho1
    tfv_d = _mf(lib, lay, lib.CELL, tfv, 1)
    dx = [1.0 / n[d] for d in range(3)]
    vol = dx[0] * dx[1] * dx[2]
    for (S_d, nc, tf_d, ic, isvel, out, start, fl) in ((Sv_d, 3, tfv_d, [mom] * 3, 1, Vs, V0, fv), (Ss_d, nscal, tfs_d, icons, 0, Ss, S0, fs)):
        scratch = lib.MultiFab(lay, lib.CELL, nc, 0)
        ed = [lib.MultiFab(lay, lib.face(d), nc, 0) for d in range(3)]
        lib.godunov_compute_aofs(g, scratch, 0, S_d, nc, tf_d, None, um_d, ic, dt, is_velocity=isvel, edge=ed)
        want = start.copy()
        for d in range(3):
            area = vol / dx[d]
            F = ed[d].gather_valid(n) * uc[d][1:-1, 1:-1, 1:-1] * area
            got_f = fl[d].gather_valid(n)
            assert np.abs(got_f - F).max() <= 1e-13 * np.abs(F).max(), (d, float(np.abs(got_f - F).max()))
            hi = [slice(None)] * 3; lo = [slice(None)] * 3; hi[d] = slice(1, None); lo[d] = slice(0, -1)
            want += (F[tuple(hi)] - F[tuple(lo)]) / vol
        got = out.gather_valid(n)
        assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), float(np.abs(got - want).max())


def test_mac_sync_compute_with_known_edge_states(gpu):
    """MacProj.cpp:733-786: Sync(sync_indx) += div(edge * Ucorr * area) / vol for one component"""
    lib = gpu
    n = (24, 16, 32)
    g = lib.Geom.make(n, prob_hi=(1.5, 1.0, 2.0))
    lay = lib.Layout.decompose(n, 8)
    add = lambda d: tuple(1 if e == d else 0 for e in range(3))
    ed = [np.stack([_smooth(n, 5 + d, add(d)), _smooth(n, 8 + d, add(d))], axis=-1) for d in range(3)]
    uc = [_smooth(n, 11 + d, add(d))[..., None] for d in range(3)]
    ed_d = [_mf(lib, lay, lib.face(d), ed[d], 0) for d in range(3)]
    uc_d = [_mf(lib, lay, lib.face(d), uc[d], 0) for d in range(3)]
    S0 = np.stack([_smooth(n, 1), _smooth(n, 2), _smooth(n, 3)], axis=-1)
    Sy = _mf(lib, lay, lib.CELL, S0, 0)
    fl = [lib.MultiFab(lay, lib.face(d), 1, 0) for d in range(3)]
    lib.mac_sync_compute_edge(g, uc_d, Sy, 2, ed_d, 1, flux=fl)
    dx = g.dx
    vol = dx[0] * dx[1] * dx[2]
    want = S0.copy()
    for d in range(3):
        F = ed[d][..., 1] * uc[d][..., 0] * (vol / dx[d])
        assert np.array_equal(fl[d].gather_valid(n)[..., 0], F)
        hi = [slice(None)] * 3; lo = [slice(None)] * 3; hi[d] = slice(1, None); lo[d] = slice(0, -1)
        want[..., 2] += (F[tuple(hi)] - F[tuple(lo)]) / vol
    got = Sy.gather_valid(n)
    assert np.array_equal(got[..., :2], S0[..., :2])
    assert np.abs(got - want).max() <= 1e-13 * np.abs(want).max()


@pytest.mark.parametrize("per", [(1, 1, 1), (1, 0, 1)])
def test_syncreg_comp_add_is_fine_add_of_the_residual_zeroed_under_the_finer_level(gpu, per):
    """SyncRegister.cpp:302-348: the nodes of the finer level's boxes (periodic images included) are zeroed in the residual, then FineAdd"""
    lib = gpu
    nc = (16, 16, 16)
    nf = tuple(2 * v for v in nc)
    cg, fg = lib.Geom.make(nc, periodic=per), lib.Geom.make(nf, periodic=per)
    clay = lib.Layout.decompose(nc, 8)
    fboxes = [((8, 8, 0), (23, 23, 15)), ((8, 8, 16), (23, 23, 31))]              # level l + 1: spans the periodic z direction
    flay = lib.Layout(fboxes)
    ffboxes = [((24, 24, 0), (39, 39, 15)), ((24, 24, 48), (39, 39, 63))]          # level l + 2 (index space 4 x coarse), touching z-lo and z-hi
    fflay = lib.Layout(ffboxes)
    rng = np.random.default_rng(3)
    R = rng.standard_normal(tuple(v + 1 for v in nf))
    for d in range(3):                  # single-valued on the periodic seams
        if per[d]:
            a = [slice(None)] * 3; b = [slice(None)] * 3; a[d] = nf[d]; b[d] = 0
            R[tuple(a)] = R[tuple(b)]
    mask = np.ones_like(R)
    for lo, hi in ffboxes:
        l2 = [lo[d] // 2 for d in range(3)]; h2 = [(hi[d] + 1) // 2 for d in range(3)]      # nodal box in the residual's index space
        for sz in (-nf[2], 0, nf[2]):
            k0, k1 = max(l2[2] + sz, 0), min(h2[2] + sz, nf[2])
            if k0 <= k1:
                mask[l2[0]:h2[0] + 1, l2[1]:h2[1] + 1, k0:k1 + 1] = 0.0
    out = []
    for use_comp_add in (True, False):
        reg = lib.SyncRegister(flay, clay, cg, fg, 2)
        rf = lib.MultiFab(flay, lib.NODE, 1, 1); rf.setval(0.0)
        G = (R if use_comp_add else R * mask)
        for li in range(rf.nlocal()):
            a, lo = rf.to_numpy(li)
            blo, bhi, _ = flay.local_box(li)
            a[1:-1, 1:-1, 1:-1, 0] = G[tuple(slice(blo[d], bhi[d] + 2) for d in range(3))]
            rf.from_numpy(a, li)
        if use_comp_add:
            reg.CompAdd(rf, fg, fflay, 2, 0.5)
            z = np.concatenate([rf.to_numpy(li)[0][1:-1, 1:-1, 1:-1, 0].ravel() for li in range(rf.nlocal())])
            assert (z == 0.0).sum() > 100                  # the residual itself is modified, as upstream's is
        else:
            reg.FineAdd(rf, 0.5)
        rhs = lib.MultiFab(clay, lib.NODE, 1, 0)
        reg.InitRHS(rhs)
        out.append(rhs.gather_valid(nc)[..., 0])
    assert np.abs(out[1]).max() > 0.1
    assert np.array_equal(out[0], out[1])


def _two_level(lib, n0=16):
    from iamr_amd import ns as NS
    from iamr_amd.amr import Amr
    g0 = lib.Geom.make((n0,) * 3)
    lo, hi = n0 // 2, n0 // 2 + n0 - 1
    lays = [lib.Layout.decompose((n0,) * 3, n0 // 2), lib.Layout([((lo, lo, lo), (hi, hi, n0 + n0 // 2 - 1)), ((lo, lo, n0 + n0 // 2), (hi, hi, hi))])]
    return NS, Amr, g0, lays


def test_initial_velocity_project_equals_the_hierarchys_post_init(gpu):
    """post_init with init_iter = 0 = initialVelocityProject + avgDown (NavierStokesBase.cpp:2369-2439): the entry on the levels' own arrays
    followed by the average gives the same velocities (to the bit when the hierarchy keeps the callers' boxes); a second projection of the
    projected field finds little left."""
    lib = gpu
    NS, Amr, g0, lays = _two_level(lib)
    fg = lib.Geom.make((32,) * 3)

    def make():
        amr = Amr(g0, lays, NS.ns_params(cfl=0.7, visc_coef=1e-3, init_iter=0, init_shrink=1.0), lib.mg_opts())
        for l in range(2):
            amr.levels[l].init_taylorgreen(1.0, 0.7, 1.3, 1.0, 0.9)      # not divergence free: something to project
        return amr
    a = make(); a.post_init()
    b = make()
    S = [b.levels[l].data(NS.NavierStokes.S_NEW) for l in range(2)]
    for l in range(2):
        S[l].fill_boundary(g0 if l == 0 else fg)
    P = [lib.MultiFab(lays[l], lib.NODE, 1, 1) for l in range(2)]
    levels = [(g0, lays[0], (0, 0, 0), (0, 0, 0), 2, None), (fg, lays[1], (0, 0, 0), (0, 0, 0), 2, None)]
    st = lib.initial_velocity_project(levels, S, [0, 0], P)
    assert st.converged >= 1
    lib.average_down(S[1], S[0], 0, 3)
    for l in range(2):
        nn = (16,) * 3 if l == 0 else (32,) * 3
        got = np.concatenate([S[l].to_numpy(li)[0][1:-1, 1:-1, 1:-1, :3].ravel() for li in range(S[l].nlocal())])
        Sa = a.levels[l].data(NS.NavierStokes.S_NEW)
        ref = np.concatenate([Sa.to_numpy(li)[0][1:-1, 1:-1, 1:-1, :3].ravel() for li in range(Sa.nlocal())])
        # (to the bit when the hierarchy keeps the callers' boxes; its merged level 0 sums in another order)
        assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max(), (l, float(np.abs(got - ref).max()))
    P2 = [lib.MultiFab(lays[l], lib.NODE, 1, 1) for l in range(2)]
    for l in range(2):
        S[l].fill_boundary(g0 if l == 0 else fg)
    st2 = lib.initial_velocity_project(levels, S, [0, 0], P2)
    # the projection is an APPROXIMATE one (cell-centred velocities, nodal pressure): projecting again finds a small, not a zero, potential
    assert st2.converged >= 1 and max(P2[l].norm0() for l in range(2)) <= 0.2 * max(P[l].norm0() for l in range(2))


def test_initial_sync_project_on_one_level_is_the_nodal_projection_of_the_acceleration(gpu):
    """Projection.cpp:970-1185 with one level: U_new <- (U_new - U_old) / dt projected with sigma = 1 / rho_half, grad phi ADDED to gp, phi
    added to P_new -- the same numbers as iamrx_nodal_projection (tests/test_gpu_ns.py: oracle) on (U_new - U_old) / dt"""
    lib = gpu
    from iamr_amd import ns as NS
    n = (32, 32, 32)
    g = lib.Geom.make(n)
    lay = lib.Layout.decompose(n, 16)
    dt = 0.02
    Un = np.stack([_smooth(n, 3 + c, ng=1) for c in range(3)] + [np.ones(tuple(v + 2 for v in n))], axis=-1)
    Uo = np.stack([_smooth(n, 13 + c, ng=1) for c in range(3)] + [np.ones(tuple(v + 2 for v in n))], axis=-1)
    rh = (1.0 + 0.3 * _smooth(n, 23, ng=1))[..., None]
    gp0 = np.stack([_smooth(n, 33 + c, ng=1) for c in range(3)], axis=-1)
    vn, vo, rho = _mf(lib, lay, lib.CELL, Un, 1), _mf(lib, lay, lib.CELL, Uo, 1), _mf(lib, lay, lib.CELL, rh, 1)
    gp = _mf(lib, lay, lib.CELL, gp0, 1)
    phi = lib.MultiFab(lay, lib.NODE, 1, 1); phi.setval(7.0)            # zeroed by the entry
    pn = lib.MultiFab(lay, lib.NODE, 1, 1); pn.setval(0.25)
    st = lib.initial_sync_project([(g, lay, (0, 0, 0), (0, 0, 0), 2, gp)], [vn], [0], [vo], [phi], [rho], dt, pres_new=[pn])
    assert st.converged >= 1
    # reference: the single-level projection entry on the acceleration
    acc = (Un - Uo) / dt
    va = _mf(lib, lay, lib.CELL, acc[..., :3], 1)
    sig = _mf(lib, lay, lib.CELL, 1.0 / rh, 1)
    ph2 = lib.MultiFab(lay, lib.NODE, 1, 1); ph2.setval(0.0)
    gp2 = _mf(lib, lay, lib.CELL, gp0, 1)
    st2 = NS.nodal_projection(g, va, 0, ph2, sig, gp=gp2, increment_gp=True)
    a, b = vn.gather_valid(n)[..., :3], va.gather_valid(n)
    assert np.abs(a - b).max() <= 1e-9 * np.abs(b).max(), float(np.abs(a - b).max())
    pa, pb = phi.gather_valid(n)[..., 0], ph2.gather_valid(n)[..., 0]
    pa = pa - pa.mean(); pb = pb - pb.mean()
    assert np.abs(pa - pb).max() <= 1e-8 * np.abs(pb).max()
    ga, gb = gp.gather_valid(n), gp2.gather_valid(n)
    assert np.abs(ga - gb).max() <= 1e-8 * np.abs(gb).max() and np.abs(ga - gp0[1:-1, 1:-1, 1:-1]).max() > 1e-3
    pnv = pn.gather_valid(n)[..., 0]
    assert np.allclose(pnv - 0.25, phi.gather_valid(n)[..., 0], rtol=0, atol=1e-13)
