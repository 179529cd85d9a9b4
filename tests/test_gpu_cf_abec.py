"""GPU parity: cell-centred multigrid on an AMR level that does not cover the domain (what MacProj::mlmg_mac_solve
and the Diffusion solves need on refined levels -- MLLinOp::setCoarseFineBC, Source/MacProj.cpp:1166-1170): Dirichlet data from
the coarse level interpolated along the coarse/fine faces (InterpBndryData, third order), applied half a coarse cell behind the
face with the level's maxorder, multigrid on the coarsened boxes.  Product (iamrx_abec_solve_cf) against the oracle
(orc_cf_interp_bndry + orc_abec_solve_cf), and against a manufactured solution."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def exact(nn, ng=1):
    x = (np.arange(-ng, nn + ng) + 0.5) / nn
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
    return np.sin(2 * np.pi * X) * np.sin(2 * np.pi * Y) * np.sin(2 * np.pi * Z) + 0.3 * np.cos(2 * np.pi * X) + 0.2 * np.sin(2 * np.pi * (Y + Z))


def setup(orc, lib, nf, boxes, variable_b, seed, maxorder):
    n, nc = (nf,) * 3, (nf // 2,) * 3
    rng = np.random.default_rng(seed)
    g_o, g_d = orc.geom(n), lib.Geom.make(n)
    gc_d = lib.Geom.make(nc)
    b_o = [orc.Fab(n, orc.face(d), 0, 1, fill=1.0) for d in range(3)]
    if variable_b:
        for d in range(3):
            b_o[d].a[...] = 0.5 + rng.random(b_o[d].a.shape)
            sl_hi = [slice(None)] * 4; sl_lo = [slice(None)] * 4
            sl_hi[d] = nf; sl_lo[d] = 0
            b_o[d].a[tuple(sl_hi)] = b_o[d].a[tuple(sl_lo)]          # periodic duplicate face
    cphi = orc.Fab(nc, orc.CELL, 1, 1)
    cphi.a[..., 0] = exact(nc[0])
    rhs = orc.Fab(n, orc.CELL, 0, 1)
    rhs.a[..., 0] = rng.standard_normal(n)
    phi0 = orc.Fab(n, orc.CELL, 1, 1)
    phi0.a[..., 0] = 0.1 * rng.standard_normal(tuple(v + 2 for v in n))
    return n, nc, g_o, g_d, gc_d, b_o, cphi, rhs, phi0


def solve_both(orc, lib, nf, boxes, variable_b=True, seed=3, maxorder=4, fixed_iters=0, rhs_override=None, rtol=1e-10, device_bottom=0):
    L = orc.lib()
    L.orc_abec_solve_cf.restype = None
    L.orc_cf_interp_bndry.restype = None
    n, nc, g_o, g_d, gc_d, b_o, cphi, rhs, phi0 = setup(orc, lib, nf, boxes, variable_b, seed, maxorder)
    if rhs_override is not None:
        rhs.a[..., 0] = rhs_override
        phi0.a[...] = 0.0
    lay = lib.Layout(boxes)
    clay = lib.Layout.single(nc)
    # product
    b_d = []
    for d in range(3):
        m = lib.MultiFab(lay, lib.face(d), 1, 0); m.set_from_global(b_o[d].a, b_o[d].lo); b_d.append(m)
    phi_d = lib.MultiFab(lay, lib.CELL, 1, 1); phi_d.set_from_global(phi0.a, phi0.lo)
    rhs_d = lib.MultiFab(lay, lib.CELL, 1, 0); rhs_d.set_from_global(rhs.a, rhs.lo)
    cphi_d = lib.MultiFab(clay, lib.CELL, 1, 1); cphi_d.set_from_global(cphi.a, cphi.lo)
    st_d = lib.abec_solve_cf(g_d, 0.0, 1.0, None, b_d, phi_d, rhs_d, cphi_d, gc_d, 2, rtol=rtol, atol=0.0,
                             opts=lib.mg_opts(maxorder=maxorder, fixed_iters=fixed_iters, device_bottom=device_bottom))
    # oracle, same multigrid depth
    Lv = orc.abec_level(g_o, b_o, boxes=boxes)
    bcv = orc.Fab(n, orc.CELL, 1, 3)
    L.orc_cf_interp_bndry(C.byref(Lv), 2, cphi.ref(), bcv.ref())
    phi = phi0.copy()
    o = orc.mg_opts(maxorder=maxorder, fixed_iters=fixed_iters, max_coarsening_level=st_d.nlevels - 1)
    st_o = orc.CMgStats()
    P = (C.c_int * 3)(0, 0, 0)
    L.orc_abec_solve_cf(C.byref(Lv), phi.ref(), rhs.ref(), P, P, bcv.ref(), C.c_double(rtol), C.c_double(0.0), C.byref(o), C.byref(st_o))
    got, ref = [], []
    for li in range(phi_d.nlocal()):
        a, lo = phi_d.to_numpy(li)
        blo, bhi, gi = lay.local_box(li)
        got.append(a[1:-1, 1:-1, 1:-1, 0])
        ref.append(phi.a[1 + blo[0]:2 + bhi[0], 1 + blo[1]:2 + bhi[1], 1 + blo[2]:2 + bhi[2], 0])
    return st_d, st_o, got, ref


CASES = {
    "one_box": [((8, 8, 8), (23, 23, 23))],
    "two_boxes_wrap_z": [((8, 8, 0), (23, 23, 15)), ((8, 8, 16), (23, 23, 31))],      # coarse/fine faces in x and y only
    "l_shape": [((0, 0, 0), (15, 15, 15)), ((16, 0, 0), (31, 15, 15)), ((0, 16, 0), (15, 31, 15))],
}


@pytest.mark.parametrize("case", list(CASES))
@pytest.mark.parametrize("maxorder", [2, 4])
def test_cf_solve_matches_oracle(orc, gpu, case, maxorder):
    boxes = CASES[case]
    st_d, st_o, got, ref = solve_both(orc, gpu, 32, boxes, maxorder=maxorder, fixed_iters=3)
    assert st_d.nlevels >= 3
    for g, r in zip(got, ref):
        assert np.abs(g - r).max() <= 1e-10 * max(1.0, np.abs(r).max()), np.abs(g - r).max()
    st_d, st_o, got, ref = solve_both(orc, gpu, 32, boxes, maxorder=maxorder)
    assert st_d.converged and st_d.iters == st_o.iters, (st_d.iters, st_o.iters)
    for g, r in zip(got, ref):
        assert np.abs(g - r).max() <= 1e-8 * max(1.0, np.abs(r).max()), np.abs(g - r).max()


@pytest.mark.parametrize("case", list(CASES))
def test_cf_solve_with_the_device_bottom_solver(orc, gpu, case):
    """the single-workgroup bottom solver on a coarse/fine level (k_abec_bottom with the CfTab ghost formula on the faces of the box that
    are not domain faces): the hierarchy ends at the first single-box level of at most 8^3 cells; same converged solution as the
    oracle's upstream-shaped hierarchy, within one cycle of its iteration count"""
    boxes = CASES[case]
    st_d, st_o, got, ref = solve_both(orc, gpu, 32, boxes, device_bottom=1)
    assert st_d.converged and abs(st_d.iters - st_o.iters) <= 1, (st_d.iters, st_o.iters)
    for g, r in zip(got, ref):
        assert np.abs(g - r).max() <= 1e-8 * max(1.0, np.abs(r).max()), np.abs(g - r).max()


def test_cf_solve_reproduces_a_smooth_solution(orc, gpu):
    """-lap(phi) = f on the refined patch with coarse data = the exact solution at the coarse cell centres: second-order accurate"""
    errs = []
    for nf in (16, 32):
        q = nf // 4
        boxes = [((q, q, q), (3 * q - 1, 3 * q - 1, 2 * q - 1)), ((q, q, 2 * q), (3 * q - 1, 3 * q - 1, 3 * q - 1))]
        x = (np.arange(nf) + 0.5) / nf
        X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
        f = 12 * np.pi ** 2 * np.sin(2 * np.pi * X) * np.sin(2 * np.pi * Y) * np.sin(2 * np.pi * Z) + 0.3 * 4 * np.pi ** 2 * np.cos(2 * np.pi * X) \
            + 0.2 * 8 * np.pi ** 2 * np.sin(2 * np.pi * (Y + Z))
        st_d, st_o, got, ref = solve_both(orc, gpu, nf, boxes, variable_b=False, maxorder=3, rhs_override=f)
        ex = exact(nf, 0)
        err = 0.0
        for g, (lo, hi) in zip(got, boxes):
            err = max(err, np.abs(g - ex[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1]).max())
        errs.append(err)
    assert errs[1] < 0.3 * errs[0] and errs[1] < 5e-3, errs


# big_patch: a 64 x 64 x 96 refined box in a 128^3 level -- the coarse/fine variants of the pair-marching colour pass and of
# the fused residual / restriction (k_abec_gsrb2<1, true, .>, k_abec_resid_restrict) run their multi-workgroup rows and 32-plane marches
@pytest.mark.parametrize("case", ["one_box", "l_shape", "big_patch"])
def test_mac_projection_on_a_refined_level(orc, gpu, case):
    """MacProj::mac_project at level 1: coarse MAC solve on the periodic 16^3 level, then the fine solve on the refined boxes with
    the coarse phi on the coarse/fine faces; variable density.  The projected fine field is discretely divergence free on the
    patch and equals the oracle's."""
    lib = gpu
    L = orc.lib()
    L.orc_mac_project_cf.restype = None
    boxes = [((32, 32, 16), (95, 95, 111))] if case == "big_patch" else CASES[case]
    nf, ncr = (128, 64) if case == "big_patch" else (32, 16)
    n, nc = (nf,) * 3, (ncr,) * 3
    rng = np.random.default_rng(5)
    g_o, g_d, gc_o, gc_d = orc.geom(n), lib.Geom.make(n), orc.geom(nc), lib.Geom.make(nc)
    dt = 0.01

    def vel_face(nn, d, seed):
        t = orc.face(d)
        ax = [(np.arange(-1, nn + t[e] + 1) + (0.0 if t[e] else 0.5)) / nn for e in range(3)]
        X, Y, Z = np.meshgrid(*ax, indexing="ij")
        ph = np.random.default_rng(seed).uniform(0, 2 * np.pi, 3)
        return np.sin(2 * np.pi * X + ph[0]) * np.cos(2 * np.pi * Y + ph[1]) + 0.5 * np.cos(2 * np.pi * Z + ph[2])

    def rho_cc(nn):
        x = (np.arange(-1, nn + 1) + 0.5) / nn
        X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
        return 1.0 + 0.3 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) + 0.1 * np.cos(4 * np.pi * Z)

    # ---- coarse level (oracle and product give the same phi to 1e-10; take the oracle's for both fine solves)
    umc = [orc.Fab(nc, orc.face(d), 1, 1) for d in range(3)]
    for d in range(3):
        umc[d].a[..., 0] = vel_face(ncr, d, 10 + d)
    rhoc = orc.Fab(nc, orc.CELL, 1, 1); rhoc.a[..., 0] = rho_cc(ncr)
    cphi = orc.Fab(nc, orc.CELL, 1, 1)
    o4 = orc.mg_opts(maxorder=4)
    st = orc.CMgStats()
    P = (C.c_int * 3)(0, 0, 0)
    L.orc_mac_project(C.byref(gc_o), orc.fabptrs(umc), rhoc.ref(), None, cphi.ref(), C.c_double(2.0 / dt), P, P,
                      C.c_double(1e-12), C.c_double(1e-16), C.byref(o4), C.byref(st))
    L.orc_fill_periodic(cphi.ref(), C.byref(gc_o), orc.i3(orc.CELL))
    # ---- fine level
    umf = [orc.Fab(n, orc.face(d), 1, 1) for d in range(3)]
    for d in range(3):
        umf[d].a[..., 0] = vel_face(nf, d, 10 + d) + 0.05 * vel_face(nf, d, 40 + d)
    rhof = orc.Fab(n, orc.CELL, 1, 1); rhof.a[..., 0] = rho_cc(nf)
    lay = lib.Layout(boxes)
    clay = lib.Layout.single(nc)
    um_d = []
    for d in range(3):
        m = lib.MultiFab(lay, lib.face(d), 1, 1); m.set_from_global(umf[d].a, umf[d].lo); um_d.append(m)
    rho_d = lib.MultiFab(lay, lib.CELL, 1, 1); rho_d.set_from_global(rhof.a, rhof.lo)
    phi_d = lib.MultiFab(lay, lib.CELL, 1, 1); phi_d.setval(0.0)
    cphi_d = lib.MultiFab(clay, lib.CELL, 1, 1); cphi_d.set_from_global(cphi.a, cphi.lo)
    st_d = lib.mlmg_mac_solve_cf(g_d, um_d, rho_d, 0, None, phi_d, 2.0 / dt, cphi_d, gc_d, 2, mac_tol=1e-11)
    assert st_d.converged
    div = lib.MultiFab(lay, lib.CELL, 1, 0)
    lib.mac_divergence(g_d, div, um_d)
    assert div.norm0() <= 1e-8
    flat = [v for lo, hi in boxes for v in (*lo, *hi)]
    bx = (C.c_int * len(flat))(*flat)
    phif = orc.Fab(n, orc.CELL, 1, 1)
    o = orc.mg_opts(maxorder=4, max_coarsening_level=st_d.nlevels - 1)
    L.orc_mac_project_cf(C.byref(g_o), orc.fabptrs(umf), rhof.ref(), None, phif.ref(), C.c_double(2.0 / dt), P, P, len(boxes), bx, 2,
                         cphi.ref(), C.c_double(1e-11), C.c_double(1e-16), C.byref(o), C.byref(st))
    assert st.iters == st_d.iters
    for li in range(phi_d.nlocal()):
        blo, bhi, gi = lay.local_box(li)
        a, lo = phi_d.to_numpy(li)
        ref = phif.a[1 + blo[0]:2 + bhi[0], 1 + blo[1]:2 + bhi[1], 1 + blo[2]:2 + bhi[2], 0]
        assert np.abs(a[1:-1, 1:-1, 1:-1, 0] - ref).max() <= 1e-8 * max(np.abs(ref).max(), 1e-3)
        for d in range(3):
            u, ulo = um_d[d].to_numpy(li)
            hi = [bhi[e] + (1 if e == d else 0) for e in range(3)]
            uref = umf[d].a[1 + blo[0]:2 + hi[0], 1 + blo[1]:2 + hi[1], 1 + blo[2]:2 + hi[2], 0]
            assert np.abs(u[1:-1, 1:-1, 1:-1, 0] - uref).max() <= 1e-8, (d, np.abs(u[1:-1, 1:-1, 1:-1, 0] - uref).max())


@pytest.mark.parametrize("nf,box", [(32, ((8, 8, 8), (23, 23, 23))), (128, ((32, 32, 16), (95, 95, 111)))])
def test_constant_masks_of_a_box_inside_the_domain(gpu, nf, box):
    """a refined level that is ONE box strictly inside the domain: the maintaining colour pass takes its coarse/fine masks as constants
    (k_abec_gsrb2<., true, true, ALLCF>) -- the same doubles as with the masks loaded (GSRB2_ALLCF = 0)"""
    lib = gpu
    n, nc = (nf,) * 3, (nf // 2,) * 3
    g_d, gc_d = lib.Geom.make(n), lib.Geom.make(nc)
    lay, clay = lib.Layout([box]), lib.Layout.single(nc)
    x = (np.arange(-1, nf + 1) + 0.5) / nf
    X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
    rho = 1.0 + 0.3 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) + 0.1 * np.cos(4 * np.pi * Z)
    out = {}
    for allcf in (0, 1):
        old = lib.tuning_get("GSRB2_ALLCF", 1)
        lib.tuning_set("GSRB2_ALLCF", allcf)
        try:
            um_d = []
            for d in range(3):
                t = lib.face(d)
                shape = tuple(nf + t[e] + 2 for e in range(3)) + (1,)
                m = lib.MultiFab(lay, t, 1, 1)
                m.set_from_global(np.random.default_rng(20 + d).standard_normal(shape), (-1, -1, -1))
                um_d.append(m)
            rho_d = lib.MultiFab(lay, lib.CELL, 1, 1); rho_d.set_from_global(rho[..., None], (-1, -1, -1))
            phi_d = lib.MultiFab(lay, lib.CELL, 1, 1); phi_d.setval(0.0)
            cphi_d = lib.MultiFab(clay, lib.CELL, 1, 1)
            cphi_d.set_from_global(exact(nc[0])[..., None], (-1, -1, -1))
            st = lib.mlmg_mac_solve_cf(g_d, um_d, rho_d, 0, None, phi_d, 200.0, cphi_d, gc_d, 2, mac_tol=1e-11)
            assert st.converged
            out[allcf] = (st.iters, phi_d.to_numpy(0)[0].copy(), [m.to_numpy(0)[0].copy() for m in um_d])
        finally:
            lib.tuning_set("GSRB2_ALLCF", old)
    assert out[0][0] == out[1][0]
    assert np.array_equal(out[0][1], out[1][1])
    for a, b in zip(out[0][2], out[1][2]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("nb", [(128, 32, 16), (256, 16, 32)])
def test_sweep_kernel_with_coarse_fine_faces_inside_a_mac_solve(gpu, nb):
    """round 5: on a refined level that is one box strictly inside its domain with 128 / 256 cells in x the finest multigrid level is smoothed
    by the one-launch sweep with in-kernel coarse/fine values (k_abec_gsrb_rb<.., W3>; directly against the oracle in
    tests/test_gpu_kernel_forms.py) instead of the maintaining colour passes, and its last sweep of a V-cycle adds the correction to the
    solution (ACC).  Each sweep forms the doubles of the two colour passes and nothing else reads the correction's ghost cells, so the whole
    MAC solve on the refined level (MacProj.cpp:1084-1184 with setCoarseFineBC, :1166-1170) is the same solve bit for bit: iterations,
    potential, projected face velocities -- under IAMRX_GSRB_RB_CF = 1 / 0 and IAMRX_MG_ACC_LAST_SWEEP = 1 / 0."""
    lib = gpu
    off = 16
    n = tuple(v + 2 * off for v in nb)
    nc = tuple(v // 2 for v in n)
    box = (tuple([off] * 3), tuple(off + v - 1 for v in nb))
    g_d, gc_d = lib.Geom.make(n, prob_hi=tuple(v / 64.0 for v in n)), lib.Geom.make(nc, prob_hi=tuple(v / 64.0 for v in n))
    lay, clay = lib.Layout([box]), lib.Layout.single(nc)
    ax = [(np.arange(-1, n[d] + 1) + 0.5) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    rho = 1.0 + 0.3 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) + 0.1 * np.cos(4 * np.pi * Z)
    axc = [(np.arange(-1, nc[d] + 1) + 0.5) / nc[d] for d in range(3)]
    Xc, Yc, Zc = np.meshgrid(*axc, indexing="ij")
    cphi = np.sin(2 * np.pi * Xc) * np.sin(2 * np.pi * Yc) * np.sin(2 * np.pi * Zc) + 0.3 * np.cos(2 * np.pi * Xc)
    out = {}
    for mode in ((1, 1), (0, 0), (1, 0)):
        old = lib.tuning_get("GSRB_RB_CF", 1), lib.tuning_get("MG_ACC_LAST_SWEEP", 1)
        lib.tuning_set("GSRB_RB_CF", mode[0]); lib.tuning_set("MG_ACC_LAST_SWEEP", mode[1])
        try:
            um_d = []
            for d in range(3):
                t = lib.face(d)
                shape = tuple(n[e] + t[e] + 2 for e in range(3)) + (1,)
                m = lib.MultiFab(lay, t, 1, 1)
                m.set_from_global(np.random.default_rng(20 + d).standard_normal(shape), (-1, -1, -1))
                um_d.append(m)
            rho_d = lib.MultiFab(lay, lib.CELL, 1, 1); rho_d.set_from_global(rho[..., None], (-1, -1, -1))
            phi_d = lib.MultiFab(lay, lib.CELL, 1, 1); phi_d.setval(0.0)
            cphi_d = lib.MultiFab(clay, lib.CELL, 1, 1); cphi_d.set_from_global(cphi[..., None], (-1, -1, -1))
            st = lib.mlmg_mac_solve_cf(g_d, um_d, rho_d, 0, None, phi_d, 200.0, cphi_d, gc_d, 2, mac_tol=1e-11)
            assert st.converged
            div = lib.MultiFab(lay, lib.CELL, 1, 0)
            lib.mac_divergence(g_d, div, um_d)
            assert div.norm0() <= 1e-7
            out[mode] = (st.iters, phi_d.to_numpy(0)[0][1:-1, 1:-1, 1:-1].copy(), [m.to_numpy(0)[0][1:-1, 1:-1, 1:-1].copy() for m in um_d])
        finally:
            lib.tuning_set("GSRB_RB_CF", old[0]); lib.tuning_set("MG_ACC_LAST_SWEEP", old[1])
    for mode in ((0, 0), (1, 0)):
        assert out[mode][0] == out[(1, 1)][0], (mode, out[mode][0], out[(1, 1)][0])
        assert np.array_equal(out[mode][1], out[(1, 1)][1]), (mode, float(np.abs(out[mode][1] - out[(1, 1)][1]).max()))
        for a, b in zip(out[mode][2], out[(1, 1)][2]):
            assert np.array_equal(a, b), mode
