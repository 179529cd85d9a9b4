"""GPU parity with physical (wall) boundaries: BC ghost fill, Godunov kernels with ext_dir / foextrap / hoextrap /
reflect BC branches, cell-centred multigrid with Neumann and Dirichlet domain BCs -- against the CPU oracle.
These are the building blocks of the LidDrivenCavity / RayleighTaylor configurations."""
import ctypes as C
import numpy as np
import pytest
from conftest import godunov_same

pytestmark = pytest.mark.gpu

# amrex::BCType
REFLECT_ODD, INT_DIR, REFLECT_EVEN, FOEXTRAP, EXT_DIR, HOEXTRAP = -1, 0, 1, 2, 3, 4


def field(n, ng, seed, typ=(0, 0, 0)):
    rng = np.random.default_rng(seed)
    ax = [(np.arange(-ng, n[d] + typ[d] + ng) + (0.0 if typ[d] else 0.5)) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    ph = rng.uniform(0, 2 * np.pi, 4)
    return np.sin(2 * np.pi * X + ph[0]) * np.cos(2 * np.pi * Y + ph[1]) + 0.5 * np.cos(2 * np.pi * Z + ph[2]) + 0.2 * np.sin(4 * np.pi * (X + Y) + ph[3])


def orc_bcrecs(orc, bcs):
    arr = (orc.CBCRec * len(bcs))()
    for n, (lo, hi) in enumerate(bcs):
        arr[n].lo = (C.c_int * 3)(*lo)
        arr[n].hi = (C.c_int * 3)(*hi)
    return arr


# per-component BCs of a box with no-slip walls in x, slip walls in z, periodic y  (velocity comps, NS_BC.H tables)
VEL_BC = [((EXT_DIR, INT_DIR, HOEXTRAP), (EXT_DIR, INT_DIR, HOEXTRAP)),
          ((EXT_DIR, INT_DIR, HOEXTRAP), (EXT_DIR, INT_DIR, HOEXTRAP)),
          ((EXT_DIR, INT_DIR, EXT_DIR), (EXT_DIR, INT_DIR, EXT_DIR))]
SCAL_BC = [((FOEXTRAP, INT_DIR, FOEXTRAP), (FOEXTRAP, INT_DIR, REFLECT_EVEN)),
           ((FOEXTRAP, INT_DIR, FOEXTRAP), (HOEXTRAP, INT_DIR, REFLECT_ODD))]
PER = (0, 1, 0)


@pytest.mark.parametrize("boxes", [None, 8])
def test_physbc_fill(orc, gpu, boxes):
    lib = gpu
    L = orc.lib()
    n = (16, 16, 16)
    g_o = orc.geom(n, periodic=PER)
    g_d = lib.Geom.make(n, periodic=PER)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    bcs = VEL_BC + SCAL_BC
    ed_lo = [[0.1 * (c + 1), 0.0, -0.2 * (c + 1)] for c in range(5)]
    ed_hi = [[0.3 * (c + 1), 0.0, 0.7 + c] for c in range(5)]
    f = orc.Fab(n, orc.CELL, 3, 5)
    for c in range(5):
        f.a[..., c] = field(n, 3, 10 + c)
    f.a[:3, :, :, :] = np.nan; f.a[-3:, :, :, :] = np.nan; f.a[:, :, :3, :] = np.nan; f.a[:, :, -3:, :] = np.nan
    m = lib.MultiFab(lay, lib.CELL, 5, 3)
    m.set_from_global(np.nan_to_num(f.a, nan=1e40), f.lo)
    f.a = np.nan_to_num(f.a, nan=1e40)
    L.orc_fill_periodic(f.ref(), C.byref(g_o), orc.i3(orc.CELL))
    el = (C.c_double * 15)(*[x for r in ed_lo for x in r])
    eh = (C.c_double * 15)(*[x for r in ed_hi for x in r])
    L.orc_fill_physbc_cc(f.ref(), C.byref(g_o), orc_bcrecs(orc, bcs), el, eh)
    m.fill_boundary(g_d)
    m.fill_physbc(g_d, bcs, ed_lo, ed_hi)
    for li in range(m.nlocal()):
        a, lo = m.to_numpy(li)
        sl = tuple(slice(lo[d] - f.lo[d], lo[d] - f.lo[d] + a.shape[d]) for d in range(3))
        assert np.array_equal(a, f.a[sl]), li


@pytest.mark.parametrize("boxes", [None, 8])
def test_godunov_with_walls(orc, gpu, boxes, scheme=0):
    lib = gpu
    L = orc.lib()
    n = (16, 16, 16)
    g_o = orc.geom(n, periodic=PER)
    g_d = lib.Geom.make(n, periodic=PER)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    ed_lo = [[0.0, 0.0, 0.0], [0.2, 0.0, 0.0], [0.0, 0.0, 0.0]]
    ed_hi = [[0.0, 0.0, 0.0], [-0.1, 0.0, 0.0], [0.0, 0.0, 0.0]]
    vel = orc.Fab(n, orc.CELL, 3, 3)
    for c in range(3):
        vel.a[..., c] = field(n, 3, 20 + c)
    L.orc_fill_periodic(vel.ref(), C.byref(g_o), orc.i3(orc.CELL))
    el = (C.c_double * 9)(*[x for r in ed_lo for x in r])
    eh = (C.c_double * 9)(*[x for r in ed_hi for x in r])
    L.orc_fill_physbc_cc(vel.ref(), C.byref(g_o), orc_bcrecs(orc, VEL_BC), el, eh)
    force = orc.Fab(n, orc.CELL, 1, 3)
    for c in range(3):
        force.a[..., c] = 2.0 * field(n, 1, 30 + c)
    dt = 0.3 / 16
    um_o = [orc.Fab(n, orc.face(d), 1, 1) for d in range(3)]
    L.orc_extrap_vel_to_faces(C.byref(g_o), vel.ref(), force.ref(), orc.fabptrs(um_o), C.c_double(dt), orc_bcrecs(orc, VEL_BC), 0)
    vel_d = lib.MultiFab(lay, lib.CELL, 3, 3); vel_d.set_from_global(vel.a, vel.lo)
    frc_d = lib.MultiFab(lay, lib.CELL, 3, 1); frc_d.set_from_global(force.a, force.lo)
    um_d = [lib.MultiFab(lay, lib.face(d), 1, 1) for d in range(3)]
    lib.godunov_extrap_vel_to_faces(g_d, vel_d, frc_d, um_d, dt, VEL_BC, 0, scheme=scheme)
    for d in range(3):
        got = um_d[d].gather_valid(n)[..., 0]
        ref = um_o[d].valid(n, orc.face(d))[..., 0]
        godunov_same(got, ref, d)
    # advection of (vel) and of two scalars with the predicted MAC velocities (ghost faces: periodic fill + copy of the wall value)
    for d in range(3):
        L.orc_fill_periodic(um_o[d].ref(), C.byref(g_o), orc.i3(orc.face(d)))
        a = um_o[d].a
        # transverse ghost faces outside the domain in non-periodic directions: first-order extrapolation (harmless, both sides identical)
        a[0, :, :, :] = a[1, :, :, :]; a[-1, :, :, :] = a[-2, :, :, :]
        a[:, :, 0, :] = a[:, :, 1, :]; a[:, :, -1, :] = a[:, :, -2, :]
        um_d[d].set_from_global(a, um_o[d].lo)
    for S_bc, ncomp, icons, isvel, seed in ((VEL_BC, 3, (0, 0, 0), 1, 40), (SCAL_BC, 2, (1, 0), 0, 50)):
        S = orc.Fab(n, orc.CELL, 3, ncomp)
        for c in range(ncomp):
            S.a[..., c] = 1.5 + field(n, 3, seed + c)
        L.orc_fill_periodic(S.ref(), C.byref(g_o), orc.i3(orc.CELL))
        edl = (C.c_double * (3 * ncomp))(*([0.25] * (3 * ncomp)))
        L.orc_fill_physbc_cc(S.ref(), C.byref(g_o), orc_bcrecs(orc, S_bc), edl, edl)
        f2 = orc.Fab(n, orc.CELL, 1, ncomp)
        aofs_o = orc.Fab(n, orc.CELL, 0, ncomp)
        edge_o = [orc.Fab(n, orc.face(d), 0, ncomp) for d in range(3)]
        ic = (C.c_int * ncomp)(*icons)
        L.orc_compute_aofs(C.byref(g_o), aofs_o.ref(), 0, S.ref(), ncomp, f2.ref(), None, orc.fabptrs(um_o), ic, C.c_double(dt),
                           orc_bcrecs(orc, S_bc), isvel, 0, orc.fabptrs(edge_o), None)
        S_d = lib.MultiFab(lay, lib.CELL, ncomp, 3); S_d.set_from_global(S.a, S.lo)
        f_d = lib.MultiFab(lay, lib.CELL, ncomp, 1); f_d.setval(0.0)
        aofs_d = lib.MultiFab(lay, lib.CELL, ncomp, 0)
        edge_d = [lib.MultiFab(lay, lib.face(d), ncomp, 0) for d in range(3)]
        lib.godunov_compute_aofs(g_d, aofs_d, 0, S_d, ncomp, f_d, None, um_d, icons, dt, S_bc, isvel, 0, edge=edge_d, scheme=scheme)
        for d in range(3):
            godunov_same(edge_d[d].gather_valid(n), edge_o[d].valid(n, orc.face(d)), ("edge", d))
        godunov_same(aofs_d.gather_valid(n), aofs_o.valid(n), "aofs")


@pytest.mark.parametrize("per,boxes", [((0, 0, 0), None), ((0, 1, 0), None), ((0, 0, 0), 8)])
def test_nodal_projection_with_neumann_walls(orc, gpu, per, boxes):
    """nodal projection in a box with solid walls (Neumann for phi; LidDrivenCavity / RayleighTaylor set-up):
    mirrored sigma and phi ghosts, zero velocity outside, doubled wall rows, half-weighted wall nodes in sums."""
    lib = gpu
    from iamr_amd import ns as N
    L = orc.lib()
    n = (16, 16, 16)
    g_o = orc.geom(n, periodic=per)
    g_d = lib.Geom.make(n, periodic=per)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    xc = [(np.arange(-1, n[d] + 1) + 0.5) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*xc, indexing="ij")
    sig = orc.Fab(n, orc.CELL, 1, 1)
    sig.a[..., 0] = 1.0 / (1.0 + 0.5 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y))
    vel = orc.Fab(n, orc.CELL, 1, 3)
    vel.a[..., 0] = np.sin(np.pi * X) * np.cos(np.pi * Y) * np.cos(np.pi * Z) + 0.2 * np.sin(np.pi * X)
    vel.a[..., 1] = np.cos(np.pi * X) * np.sin(2 * np.pi * Y) * np.cos(np.pi * Z)
    vel.a[..., 2] = 0.7 * np.cos(np.pi * X) * np.cos(np.pi * Y) * np.sin(np.pi * Z)
    L.orc_fill_periodic(vel.ref(), C.byref(g_o), orc.i3(orc.CELL))
    bc = [0 if per[d] else 102 for d in range(3)]
    vel_d = lib.MultiFab(lay, lib.CELL, 3, 1); vel_d.set_from_global(vel.a, vel.lo)
    sig_d = lib.MultiFab(lay, lib.CELL, 1, 1); sig_d.set_from_global(sig.a, sig.lo)
    p_o = orc.Fab(n, orc.NODE, 1, 1)
    st_o = orc.CMgStats()
    oo = orc.mg_opts()
    L.orc_nodal_project(C.byref(g_o), vel.ref(), p_o.ref(), sig.ref(), orc.i3(bc), orc.i3(bc), C.c_double(1e-11), C.c_double(1e-16),
                        C.byref(oo), C.byref(st_o))
    p_d = lib.MultiFab(lay, lib.NODE, 1, 1); p_d.setval(0.0)
    st = N.nodal_projection(g_d, vel_d, 0, p_d, sig_d, 0, lobc=bc, hibc=bc, rel_tol=1e-11, opts=lib.mg_opts(**orc.UPSTREAM_NODAL_CYCLE))
    assert st.converged == 1 and st_o.converged == 1 and st.iters == st_o.iters, (st.iters, st_o.iters)
    pg = p_d.gather_valid(n)[..., 0]; pr = p_o.valid(n, orc.NODE)[..., 0]
    assert np.abs((pg - pg.mean()) - (pr - pr.mean())).max() <= 1e-8 * np.abs(pr - pr.mean()).max()
    assert np.abs(vel_d.gather_valid(n) - vel.valid(n)).max() <= 1e-8


@pytest.mark.parametrize("per", [(0, 0, 0), (0, 1, 0)])
def test_tensor_operator_with_noslip_walls(orc, gpu, per):
    """MLTensorOp with Dirichlet (no-slip / moving lid) velocity BCs: face extrapolation (max_order 2), edge/corner ghost
    cells for the cross terms, explicit apply and Crank-Nicolson solve -- the LidDrivenCavity viscous step."""
    lib = gpu
    from iamr_amd import ns as N
    L = orc.lib()
    n = (16, 16, 16)
    g_o = orc.geom(n, periodic=per)
    g_d = lib.Geom.make(n, periodic=per)
    lay = lib.Layout.single(n)
    bc = [0 if per[d] else 101 for d in range(3)]
    xc = [(np.arange(-1, n[d] + 1) + 0.5) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*xc, indexing="ij")
    u = orc.Fab(n, orc.CELL, 1, 3)
    u.a[..., 0] = np.sin(np.pi * X) * np.cos(2 * np.pi * Y) * np.sin(np.pi * Z) + 0.3 * Z
    u.a[..., 1] = np.cos(3 * X) * np.sin(2 * np.pi * Y) + 0.2 * X * Z
    u.a[..., 2] = 0.5 * np.sin(2 * np.pi * (X + Y)) * Z * (1 - Z)
    L.orc_fill_periodic(u.ref(), C.byref(g_o), orc.i3(orc.CELL))
    # boundary values live in the ghost cells (incl. edges/corners): lid u = 1 at z-hi, zero elsewhere
    if not per[0]:
        u.a[0, :, :, :] = 0.0; u.a[-1, :, :, :] = 0.0
    if not per[1]:
        u.a[:, 0, :, :] = 0.0; u.a[:, -1, :, :] = 0.0
    u.a[:, :, 0, :] = 0.0
    u.a[:, :, -1, :] = 0.0
    u.a[:, :, -1, 0] = 1.0
    eta_o, eta_d = [], []
    for d in range(3):
        e = orc.Fab(n, orc.face(d), 0, 1, fill=0.01)
        eta_o.append(e)
        m = lib.MultiFab(lay, lib.face(d), 1, 0); m.setval(0.01); eta_d.append(m)
    u_d = lib.MultiFab(lay, lib.CELL, 3, 1); u_d.set_from_global(u.a, u.lo)
    # explicit apply (a = 0, b = -1): needs the BC ghost fill incl. edges -> go through the oracle's solver BC path
    lev_b = []
    for d in range(3):
        b3 = orc.Fab(n, orc.face(d), 0, 3)
        for c in range(3):
            b3.a[..., c] = 0.01 * (4.0 / 3.0 if c == d else 1.0)
        lev_b.append(b3)
    lev = orc.abec_level(g_o, lev_b, alpha=0.0, beta=-1.0, ncomp=3, tensor=1)
    uo = u.copy()
    L.orc_abec_applybc(C.byref(lev), uo.ref(), orc.i3(bc), orc.i3(bc), 2, 1, u.ref())
    y = orc.Fab(n, orc.CELL, 0, 3)
    L.orc_abec_apply(C.byref(lev), y.ref(), uo.ref())
    out_d = lib.MultiFab(lay, lib.CELL, 3, 0)
    N.tensor_apply(g_d, out_d, u_d, 0.0, -1.0, None, eta_d, lobc=bc, hibc=bc, maxorder=2)
    assert np.abs(out_d.gather_valid(n) - y.a).max() <= 1e-11 * np.abs(y.a).max()
    ug, _ = u_d.to_numpy(0)
    assert np.abs(ug - uo.a).max() <= 1e-13      # ghost cells incl. edges and corners agree
    # implicit solve (rho - theta dt div tau) u = rhs
    acoef = orc.Fab(n, orc.CELL, 0, 1, fill=1.0)
    rhs = orc.Fab(n, orc.CELL, 0, 3)
    rhs.a[...] = u.valid(n)
    s_o = u.copy()
    st_o = orc.CMgStats()
    oo = orc.mg_opts(maxorder=2)
    L.orc_tensor_solve(C.byref(g_o), s_o.ref(), rhs.ref(), C.c_double(1.0), C.c_double(0.02), acoef.ref(), orc.fabptrs(eta_o),
                       orc.i3(bc), orc.i3(bc), C.c_double(1e-10), C.c_double(0.0), C.byref(oo), C.byref(st_o))
    a_d = lib.MultiFab(lay, lib.CELL, 1, 0); a_d.setval(1.0)
    r_d = lib.MultiFab(lay, lib.CELL, 3, 0); r_d.set_from_global(rhs.a, rhs.lo)
    s_d = lib.MultiFab(lay, lib.CELL, 3, 1); s_d.set_from_global(u.a, u.lo)
    st = N.tensor_solve(g_d, s_d, r_d, 1.0, 0.02, a_d, eta_d, lobc=bc, hibc=bc, tol_rel=1e-10, tol_abs=0.0, opts=lib.mg_opts(maxorder=2))
    assert st.converged == 1 and st.iters == st_o.iters
    assert np.abs(s_d.gather_valid(n) - s_o.valid(n)).max() <= 1e-8


@pytest.mark.parametrize("bctype,alpha", [(102, 0.0), (101, 0.0), (101, 1.0), (102, 1.0)])
def test_cell_mg_with_domain_bcs(orc, gpu, bctype, alpha):
    """MLABecLaplacian solve with Neumann (102, MAC projection at walls) / Dirichlet (101, diffusion) faces in x and z,
    periodic y; inhomogeneous Dirichlet data in the ghost cells; max_order 2 and 3."""
    lib = gpu
    L = orc.lib()
    n = (16, 16, 16)
    g_o = orc.geom(n, periodic=PER)
    g_d = lib.Geom.make(n, periodic=PER)
    lay = lib.Layout.single(n)
    lobc = (bctype, 0, bctype)
    hibc = (bctype, 0, 102 if bctype == 102 else 101)
    b_o, b_d = [], []
    for d in range(3):
        t = orc.face(d)
        bf = orc.Fab(n, t, 0, 1)
        bf.a[..., 0] = 1.0 + 0.3 * field(n, 0, 60 + d, t) ** 2
        b_o.append(bf)
        m = lib.MultiFab(lay, t, 1, 0); m.set_from_global(bf.a, bf.lo); b_d.append(m)
    acoef = orc.Fab(n, orc.CELL, 0, 1)
    acoef.a[..., 0] = 1.0 + 0.2 * field(n, 0, 70) ** 2
    a_d = lib.MultiFab(lay, lib.CELL, 1, 0); a_d.set_from_global(acoef.a, acoef.lo)
    for maxorder in (2, 3):
        phi = orc.Fab(n, orc.CELL, 1, 1)
        phi.a[..., 0] = 0.1 * field(n, 1, 80)          # initial guess; ghost cells = Dirichlet boundary data
        rhs = orc.Fab(n, orc.CELL, 0, 1)
        rhs.a[..., 0] = field(n, 0, 90)
        if bctype == 102 and alpha == 0.0:
            rhs.a[...] -= rhs.a.mean()
        lev = orc.abec_level(g_o, b_o, alpha=alpha, beta=0.7, a=acoef if alpha else None)
        st_o = orc.CMgStats()
        oo = orc.mg_opts(maxorder=maxorder)
        phi0 = phi.a.copy()
        phi_d = lib.MultiFab(lay, lib.CELL, 1, 1); phi_d.set_from_global(phi.a, phi.lo)
        rhs_d = lib.MultiFab(lay, lib.CELL, 1, 0); rhs_d.set_from_global(rhs.a, rhs.lo)
        L.orc_abec_solve(C.byref(lev), phi.ref(), rhs.ref(), orc.i3(lobc), orc.i3(hibc), C.c_double(1e-11), C.c_double(0.0), C.byref(oo), C.byref(st_o))
        ref = phi.valid(n)[..., 0]
        if bctype == 102 and alpha == 0.0:
            ref = ref - ref.mean()
        # device_bottom = 0: the oracle's hierarchy (coarsened to 2^3, host-driven BiCGStab) -> the same iteration count;
        # default: the hierarchy ends at 8^3 with the single-workgroup device bottom solver -> the same solution
        for device_bottom in (0, 1):
            phi_d.set_from_global(phi0, phi.lo)
            st = lib.abec_solve(g_d, alpha, 0.7, a_d if alpha else None, b_d, phi_d, rhs_d, lobc, hibc, rtol=1e-11, atol=0.0,
                                opts=lib.mg_opts(maxorder=maxorder, device_bottom=device_bottom))
            assert st.converged == 1 and st_o.converged == 1
            if device_bottom == 0:
                assert st.iters == st_o.iters, (st.iters, st_o.iters)
            else:
                assert st.iters <= st_o.iters + 1 and st.nlevels == 2, (st.iters, st_o.iters, st.nlevels)
            got = phi_d.gather_valid(n)[..., 0]
            if bctype == 102 and alpha == 0.0:
                got = got - got.mean()
            assert np.abs(got - ref).max() <= 1e-8 * max(np.abs(ref).max(), 1e-3)


def test_tensor_solve_slip_walls_per_component_bc(orc, gpu):
    """Slip walls (LidDrivenCavity x-lo/y-lo): the wall-normal velocity component is Dirichlet, the tangential ones
    Neumann -- MLTensorOp gets one BC triple per component (reference Source/Diffusion.cpp:724-731, 1939-2020);
    z-hi is the moving lid (all Dirichlet)."""
    lib = gpu
    from iamr_amd import ns as N
    L = orc.lib()
    n = (16, 16, 16)
    per = (0, 0, 0)
    g_o = orc.geom(n, periodic=per)
    g_d = lib.Geom.make(n, periodic=per)
    lay = lib.Layout.single(n)
    D, Nm = 101, 102
    # comp n, direction d: x-lo / y-lo slip, everything else no-slip
    lobc = [[D if (d == 2 or d == c) else Nm for d in range(3)] for c in range(3)]
    hibc = [[D, D, D] for c in range(3)]
    rng = np.random.default_rng(5)
    u = orc.Fab(n, orc.CELL, 1, 3)
    u.a[...] = 0.0
    u.a[1:-1, 1:-1, 1:-1, :] = rng.standard_normal(tuple(n) + (3,))
    u.a[:, :, -1, 0] = 1.0     # lid
    eta_o, eta_d = [], []
    for d in range(3):
        e = orc.Fab(n, orc.face(d), 0, 1)
        e.a[...] = 0.01 * (1.0 + 0.5 * rng.random(e.a.shape))
        eta_o.append(e)
        m = lib.MultiFab(lay, lib.face(d), 1, 0); m.set_from_global(e.a, e.lo); eta_d.append(m)
    acoef = orc.Fab(n, orc.CELL, 0, 1, fill=1.0)
    rhs = orc.Fab(n, orc.CELL, 0, 3)
    rhs.a[...] = u.valid(n)
    s_o = u.copy()
    st_o = orc.CMgStats()
    oo = orc.mg_opts(maxorder=2)
    lo9 = (C.c_int * 9)(*[v for c in lobc for v in c])
    hi9 = (C.c_int * 9)(*[v for c in hibc for v in c])
    L.orc_tensor_solve_bcn(C.byref(g_o), s_o.ref(), rhs.ref(), C.c_double(1.0), C.c_double(0.05), acoef.ref(), orc.fabptrs(eta_o),
                           lo9, hi9, C.c_double(1e-10), C.c_double(0.0), C.byref(oo), C.byref(st_o))
    a_d = lib.MultiFab(lay, lib.CELL, 1, 0); a_d.setval(1.0)
    r_d = lib.MultiFab(lay, lib.CELL, 3, 0); r_d.set_from_global(rhs.a, rhs.lo)
    s_d = lib.MultiFab(lay, lib.CELL, 3, 1); s_d.set_from_global(u.a, u.lo)
    st = N.tensor_solve(g_d, s_d, r_d, 1.0, 0.05, a_d, eta_d, lobc=lobc, hibc=hibc, tol_rel=1e-10, tol_abs=0.0,
                        opts=lib.mg_opts(maxorder=2))
    assert st_o.converged == 1
    assert st.converged == 1 and st.iters == st_o.iters
    assert np.abs(s_d.gather_valid(n) - s_o.valid(n)).max() <= 1e-8
    # the per-component result differs from the all-Dirichlet one (the test would otherwise not exercise anything)
    s_o2 = u.copy()
    L.orc_tensor_solve(C.byref(g_o), s_o2.ref(), rhs.ref(), C.c_double(1.0), C.c_double(0.05), acoef.ref(), orc.fabptrs(eta_o),
                       orc.i3((D, D, D)), orc.i3((D, D, D)), C.c_double(1e-10), C.c_double(0.0), C.byref(oo), C.byref(st_o))
    assert np.abs(s_o2.valid(n) - s_o.valid(n)).max() > 1e-4
