"""GPU parity: nodal solves with Dirichlet nodes -- (a) Dirichlet (outflow) domain faces, SURVEY row f3, and (b) a level that
does not cover the domain (AMR level > 0: the nodes on its boundary inside the domain hold Dirichlet data) --
through the C-ABI (iamrx_nodal_solve) against the oracle (orc_nodal_solve_cov).  Same multigrid on both sides (8-colour GS,
full-weighting restriction, sigma-weighted interpolation, BiCGStab bottom); results agree to round-off of the different
summation orders (tolerance stated per assert)."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PERIODIC, DIRICHLET, NEUMANN = 0, 101, 102


def setup_fields(orc, n, seed):
    rng = np.random.default_rng(seed)
    sig = orc.Fab(n, orc.CELL, 1, 1)
    sig.a[...] = 0.5 + rng.random(sig.a.shape)
    rhs = orc.Fab(n, orc.NODE, 0, 1)
    rhs.a[...] = rng.standard_normal(rhs.a.shape)
    phi = orc.Fab(n, orc.NODE, 1, 1)
    phi.a[...] = rng.standard_normal(phi.a.shape)          # also the Dirichlet data
    return sig, rhs, phi


def make_periodic_consistent(a, n, per, node=True):
    """the periodic duplicate node n equals node 0 (array index: ghost offset 1)"""
    for d in range(3):
        if per[d] and node:
            sl_hi = [slice(None)] * 4; sl_lo = [slice(None)] * 4
            off = 1 if a.shape[d] == n[d] + 3 else 0
            sl_hi[d] = off + n[d]; sl_lo[d] = off
            a[tuple(sl_hi)] = a[tuple(sl_lo)]


def run_both(orc, lib, n, per, lobc, hibc, boxes, owners_cov, seed, fixed_iters=0, rtol=1e-10, sigma_const=None):
    from iamr_amd import ns as N
    L = orc.lib()
    L.orc_nodal_solve_cov.restype = None
    g_o = orc.geom(n, periodic=per)
    g_d = lib.Geom.make(n, periodic=per)
    sig, rhs, phi = setup_fields(orc, n, seed)
    if sigma_const is not None:
        sig.a[...] = sigma_const
    make_periodic_consistent(rhs.a, n, per)
    make_periodic_consistent(phi.a, n, per)
    lay = lib.Layout(boxes)
    cov = None
    if owners_cov:
        cov = orc.Fab(n, orc.CELL, 0, 1)
        for lo, hi in boxes:
            cov.a[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1, 0] = 1.0
    # product
    sig_d = lib.MultiFab(lay, lib.CELL, 1, 1); sig_d.set_from_global(sig.a, sig.lo)
    rhs_d = lib.MultiFab(lay, lib.NODE, 1, 0); rhs_d.set_from_global(rhs.a, rhs.lo)
    phi_d = lib.MultiFab(lay, lib.NODE, 1, 1); phi_d.set_from_global(phi.a, phi.lo)
    st_d = N.nodal_solve(g_d, phi_d, rhs_d, sig_d, 0, lobc, hibc, rtol, 0.0, lib.mg_opts(fixed_iters=fixed_iters, **orc.UPSTREAM_NODAL_CYCLE))
    # oracle with the same multigrid depth
    o = orc.mg_opts(fixed_iters=fixed_iters, max_coarsening_level=st_d.nlevels - 1)
    st_o = orc.CMgStats()
    L.orc_nodal_solve_cov(C.byref(g_o), phi.ref(), rhs.ref(), sig.ref(), orc.i3(lobc), orc.i3(hibc), cov.ref() if cov else None,
                          C.c_double(rtol), C.c_double(0.0), C.byref(o), C.byref(st_o))
    got = []
    ref = []
    for li in range(phi_d.nlocal()):
        a, lo = phi_d.to_numpy(li)
        blo, bhi, gi = lay.local_box(li)
        v = a[1:-1, 1:-1, 1:-1, 0]                          # valid nodes blo .. bhi+1
        got.append(v)
        ref.append(phi.a[1 + blo[0]:1 + bhi[0] + 2, 1 + blo[1]:1 + bhi[1] + 2, 1 + blo[2]:1 + bhi[2] + 2, 0])
    return st_d, st_o, got, ref


@pytest.mark.parametrize("case", ["outflow_x", "outflow_all", "outflow_two_boxes"])
def test_dirichlet_faces(orc, gpu, case):
    lib = gpu
    if case == "outflow_x":         # channel: inflow/walls Neumann, outflow Dirichlet at x-hi, periodic in y
        n, per = (32, 16, 16), (0, 1, 0)
        lobc, hibc = (NEUMANN, PERIODIC, NEUMANN), (DIRICHLET, PERIODIC, NEUMANN)
        boxes = [((0, 0, 0), (31, 15, 15))]
    elif case == "outflow_all":
        n, per = (16, 16, 16), (0, 0, 0)
        lobc, hibc = (DIRICHLET,) * 3, (DIRICHLET,) * 3
        boxes = [((0, 0, 0), (15, 15, 15))]
    else:
        n, per = (32, 16, 16), (0, 0, 0)
        lobc, hibc = (NEUMANN, NEUMANN, DIRICHLET), (DIRICHLET, NEUMANN, NEUMANN)
        boxes = [((0, 0, 0), (15, 15, 15)), ((16, 0, 0), (31, 15, 15))]
    # same number of V-cycles on both sides: the iterates agree to round-off
    st_d, st_o, got, ref = run_both(orc, lib, n, per, lobc, hibc, boxes, False, 7, fixed_iters=3)
    for g, r in zip(got, ref):
        assert np.abs(g - r).max() <= 1e-10 * max(1.0, np.abs(r).max()), np.abs(g - r).max()
    # to convergence: same iteration count, same answer
    st_d, st_o, got, ref = run_both(orc, lib, n, per, lobc, hibc, boxes, False, 7)
    assert st_d.converged and st_d.iters == st_o.iters, (st_d.iters, st_o.iters)
    for g, r in zip(got, ref):
        assert np.abs(g - r).max() <= 1e-9 * max(1.0, np.abs(r).max()), np.abs(g - r).max()


@pytest.mark.parametrize("case", ["one_box", "two_boxes_periodic_z", "l_shape"])
def test_level_not_covering_the_domain(orc, gpu, case):
    lib = gpu
    n, per = (32, 32, 32), (1, 1, 1)
    P = (PERIODIC,) * 3
    if case == "one_box":
        boxes = [((8, 8, 8), (23, 23, 23))]
    elif case == "two_boxes_periodic_z":     # the level wraps around the periodic z direction: coarse/fine boundary in x and y only
        boxes = [((8, 8, 0), (23, 23, 15)), ((8, 8, 16), (23, 23, 31))]
    else:
        boxes = [((0, 0, 0), (15, 15, 15)), ((16, 0, 0), (31, 15, 15)), ((0, 16, 0), (15, 31, 15))]
    st_d, st_o, got, ref = run_both(orc, lib, n, per, P, P, boxes, True, 11, fixed_iters=3)
    assert st_d.nlevels >= 3
    for g, r in zip(got, ref):
        assert np.abs(g - r).max() <= 1e-10 * max(1.0, np.abs(r).max()), np.abs(g - r).max()
    st_d, st_o, got, ref = run_both(orc, lib, n, per, P, P, boxes, True, 11)
    assert st_d.converged and st_d.iters == st_o.iters, (st_d.iters, st_o.iters)
    for g, r in zip(got, ref):
        assert np.abs(g - r).max() <= 1e-9 * max(1.0, np.abs(r).max()), np.abs(g - r).max()


@pytest.mark.parametrize("case", ["one_box", "l_shape"])
def test_nodal_projection_on_a_refined_level(orc, gpu, case):
    """Projection::doMLMGNodalProjection for one AMR level > 0 (single-level solve, nodes on the coarse/fine boundary Dirichlet):
    product iamrx_nodal_projection on a layout that does not cover the domain against orc_nodal_project_cov; velocity
    and iteration count agree (the projection is approximate: D(u - sig G phi) is not zero, so there is no divergence pin)"""
    from iamr_amd import ns as N
    lib = gpu
    L = orc.lib()
    L.orc_nodal_project_cov.restype = None
    n, per = (32, 32, 32), (1, 1, 1)
    P = (PERIODIC,) * 3
    boxes = [((8, 8, 8), (23, 23, 23))] if case == "one_box" else \
        [((0, 0, 0), (15, 15, 15)), ((16, 0, 0), (31, 15, 15)), ((0, 16, 0), (15, 31, 15))]
    g_o, g_d = orc.geom(n, periodic=per), lib.Geom.make(n, periodic=per)
    rng = np.random.default_rng(21)
    x = (np.arange(-1, n[0] + 1) + 0.5) / n[0]
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
    vel = orc.Fab(n, orc.CELL, 1, 3)
    vel.a[..., 0] = np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) + 0.3 * np.cos(2 * np.pi * Z)
    vel.a[..., 1] = np.cos(2 * np.pi * X) * np.sin(4 * np.pi * Y) + 0.2 * np.sin(2 * np.pi * Z)
    vel.a[..., 2] = 0.5 * np.sin(2 * np.pi * (X + Z)) * np.cos(2 * np.pi * Y)
    sig = orc.Fab(n, orc.CELL, 1, 1)
    sig.a[..., 0] = 1.0 / (1.0 + 0.3 * np.sin(2 * np.pi * X) * np.sin(2 * np.pi * Y))
    phi = orc.Fab(n, orc.NODE, 1, 1)            # zero Dirichlet data on the coarse/fine boundary (incremental projection)
    cov = orc.Fab(n, orc.CELL, 0, 1)
    for lo, hi in boxes:
        cov.a[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1, 0] = 1.0
    lay = lib.Layout(boxes)
    vel_d = lib.MultiFab(lay, lib.CELL, 3, 1); vel_d.set_from_global(vel.a, vel.lo)
    sig_d = lib.MultiFab(lay, lib.CELL, 1, 1); sig_d.set_from_global(sig.a, sig.lo)
    phi_d = lib.MultiFab(lay, lib.NODE, 1, 1); phi_d.setval(0.0)
    st_d = N.nodal_projection(g_d, vel_d, 0, phi_d, sig_d, 0, P, P, 1e-11, 0.0, opts=lib.mg_opts(**orc.UPSTREAM_NODAL_CYCLE))
    o = orc.mg_opts(max_coarsening_level=st_d.nlevels - 1)
    st_o = orc.CMgStats()
    L.orc_nodal_project_cov(C.byref(g_o), vel.ref(), phi.ref(), sig.ref(), orc.i3(P), orc.i3(P), cov.ref(), C.c_double(1e-11), C.c_double(0.0),
                            C.byref(o), C.byref(st_o))
    assert st_d.converged and st_d.iters == st_o.iters
    for li in range(vel_d.nlocal()):
        blo, bhi, gi = lay.local_box(li)
        a, lo = vel_d.to_numpy(li)
        ref = vel.a[1 + blo[0]:2 + bhi[0], 1 + blo[1]:2 + bhi[1], 1 + blo[2]:2 + bhi[2], :]
        assert np.abs(a[1:-1, 1:-1, 1:-1, :] - ref).max() <= 1e-9


GSR_CASES = {
    # sizes at which NodalMG selects the register-resident plane-fused pass k_nodal_gsr on the finest level (boxes >= 48 cells in x and y;
    # VERDICT round 4, weak 1a): index wrap on a periodic box, ghost-filled boxes, walls, an outflow face (masked variant), a patch of a
    # refined level (masked), constant sigma
    "wrap 96x64x32": dict(n=(96, 64, 32), per=(1, 1, 1), lobc=(PERIODIC,) * 3, hibc=(PERIODIC,) * 3, boxes=[((0, 0, 0), (95, 63, 31))]),
    "wrap, constant sigma 64x64x32": dict(n=(64, 64, 32), per=(1, 1, 1), lobc=(PERIODIC,) * 3, hibc=(PERIODIC,) * 3, boxes=[((0, 0, 0), (63, 63, 31))], sigma_const=0.75),
    "two ghost-filled boxes 128x64x32": dict(n=(128, 64, 32), per=(1, 1, 1), lobc=(PERIODIC,) * 3, hibc=(PERIODIC,) * 3,
                                             boxes=[((0, 0, 0), (63, 63, 31)), ((64, 0, 0), (127, 63, 31))]),
    "walls 64x96x16": dict(n=(64, 96, 16), per=(0, 0, 0), lobc=(NEUMANN,) * 3, hibc=(NEUMANN,) * 3, boxes=[((0, 0, 0), (63, 95, 15))]),
    "outflow (masked) 64x64x32": dict(n=(64, 64, 32), per=(0, 1, 0), lobc=(NEUMANN, PERIODIC, NEUMANN), hibc=(DIRICHLET, PERIODIC, NEUMANN),
                                      boxes=[((0, 0, 0), (63, 63, 31))]),
    "refined patch (masked) 64x64x16 in 128x128x32": dict(n=(128, 128, 32), per=(1, 1, 1), lobc=(PERIODIC,) * 3, hibc=(PERIODIC,) * 3,
                                                          boxes=[((24, 32, 8), (87, 95, 23))], cover=True),
}


@pytest.mark.parametrize("case", list(GSR_CASES))
def test_register_resident_pass_inside_the_solver_matches_the_oracle(orc, gpu, case):
    """one and two V-cycles of iamrx_nodal_solve in the oracle's cycle shape (4 Gauss-Seidel sweeps per smooth call, 2 + 2 calls per level)
    against orc_nodal_solve_cov with the same number of cycles: every variant of k_nodal_gsr under the oracle at a size where it is the
    kernel that runs.  Tolerance 1e-10 of the iterate's size: the 27-point update is summed in another order than the oracle's
    element-by-element assembly, 16 sweeps per level and cycle."""
    lib = gpu
    c = GSR_CASES[case]
    assert lib.tuning_get("GSR", 1) != 0 and lib.tuning_get("GSR_MIN", 48) <= 48
    for iters in (1, 2):
        st_d, st_o, got, ref = run_both(orc, lib, c["n"], c["per"], c["lobc"], c["hibc"], c["boxes"], bool(c.get("cover")), 23, fixed_iters=iters,
                                        sigma_const=c.get("sigma_const"))
        for g, r in zip(got, ref):
            assert np.abs(g - r).max() <= 1e-10 * max(1.0, np.abs(r).max()), (iters, float(np.abs(g - r).max()))
