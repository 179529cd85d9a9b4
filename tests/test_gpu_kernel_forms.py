"""GPU parity, directly against the oracle, of the finest-level kernel FORMS that carry the headline since round 3:
the pair-marching colour pass k_abec_gsrb2, the pair-marching residual and the fused residual + restriction k_abec_resid_restrict -- in the
`sig` mode (MAC operator, face coefficients recomputed from the cell-centred density) and the `b_uniform` mode (three constants) -- and the
fused tensor residual k_tensor_cross_zm<.., FUSE>, at 64 x 64 x 96 and beyond: sizes where a row takes several workgroups, a thread marches 32
planes (GSRB2_TZ) and the XCD-aware tile order is on.  Through the C-ABI (iamrx_abec_form, iamrx_tensor_apply); single box, several boxes, and
the index wrap of a periodic single box.  Bit-exact for the cell-centred kernels (same expression order as oracle/orc_abec.c, contraction off
on both sides), 1e-12 relative for the tensor operator (different but equivalent summation order)."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def fields(n, seed):
    rng = np.random.default_rng(seed)
    ax = [(np.arange(-1, n[d] + 1) + 0.5) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    rho = 1.0 + 0.4 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) * np.cos(4 * np.pi * Z) + 0.05 * rng.random(X.shape)
    phi = np.sin(2 * np.pi * X) * np.cos(4 * np.pi * Y) + 0.5 * np.cos(2 * np.pi * (Z + X)) + 0.1 * rng.standard_normal(X.shape)
    rhs = (np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Z) + 0.1 * rng.standard_normal(X.shape))[1:-1, 1:-1, 1:-1]

    def wrap(a):                       # periodic ghost cells
        for d in range(3):
            lo = [slice(None)] * 3; hi = [slice(None)] * 3; s0 = [slice(None)] * 3; s1 = [slice(None)] * 3
            lo[d] = 0; s0[d] = n[d]; hi[d] = n[d] + 1; s1[d] = 1
            a[tuple(lo)] = a[tuple(s0)]; a[tuple(hi)] = a[tuple(s1)]
        return a
    return wrap(rho), wrap(phi), rhs


@pytest.mark.parametrize("coef", [1, 2])
@pytest.mark.parametrize("n,boxes", [((64, 64, 96), None), ((64, 64, 96), (32, 32, 48)), ((128, 96, 64), (64, 48, 64)), ((96, 80, 72), None)])
def test_colour_pass_residual_and_restriction_match_the_oracle(orc, gpu, n, boxes, coef):
    lib = gpu
    L = orc.lib()
    g_o, g_d = orc.geom(n), lib.Geom.make(n)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    rho, phi, rhs = fields(n, 7)
    scale, bu, beta = 0.37, (0.8, 1.1, 1.3), 1.0
    # the oracle takes face arrays: b_d = scale / (0.5 (rho(cell - e_d) + rho(cell)))  [MacProj.cpp:1098-1128's coefficients] or constants
    b_o = []
    for d in range(3):
        bf = orc.Fab(n, orc.face(d), 0, 1)
        if coef == 1:
            lo = [slice(1, n[e] + 1) for e in range(3)]; hi = [slice(1, n[e] + 1) for e in range(3)]
            lo[d] = slice(0, n[d] + 1); hi[d] = slice(1, n[d] + 2)
            bf.a[..., 0] = scale / (0.5 * (rho[tuple(lo)] + rho[tuple(hi)]))
        else:
            bf.a[...] = bu[d]
        b_o.append(bf)
    lev = orc.abec_level(g_o, b_o, beta=beta)
    phi_o = orc.Fab(n, orc.CELL, 1, 1); phi_o.a[..., 0] = phi
    rhs_o = orc.Fab(n, orc.CELL, 0, 1); rhs_o.a[..., 0] = rhs
    rho_d = lib.MultiFab(lay, lib.CELL, 1, 1); rho_d.set_from_global(rho[..., None], (-1,) * 3)
    phi_d = lib.MultiFab(lay, lib.CELL, 1, 1); phi_d.set_from_global(phi[..., None], (-1,) * 3)
    rhs_d = lib.MultiFab(lay, lib.CELL, 1, 0); rhs_d.set_from_global(rhs[..., None], (0,) * 3)
    kw = dict(rho=rho_d, scale=scale, bu=bu, beta=beta)
    z3 = orc.i3([0, 0, 0])
    single = boxes is None
    # two full sweeps; on a single box the second one through the index-wrap form (no ghost cell is read: poison them)
    for sweep in range(2):
        for rb in (0, 1):
            L.orc_fill_periodic(phi_o.ref(), C.byref(g_o), orc.i3(orc.CELL))
            L.orc_abec_gsrb(C.byref(lev), phi_o.ref(), rhs_o.ref(), rb, C.c_double(1.15), z3, z3, 3)
            if single and sweep == 1:
                lib.abec_form(g_d, coef, 4 + rb, phi_d, rhs_d, **kw)
            else:
                phi_d.fill_boundary(g_d)
                lib.abec_form(g_d, coef, rb, phi_d, rhs_d, **kw)
            got, ref = phi_d.gather_valid(n), phi_o.valid(n)
            assert np.array_equal(got, ref), ("colour pass", sweep, rb, float(np.abs(got - ref).max()))
    # residual
    L.orc_fill_periodic(phi_o.ref(), C.byref(g_o), orc.i3(orc.CELL))
    y = orc.Fab(n, orc.CELL, 0, 1)
    L.orc_abec_apply(C.byref(lev), y.ref(), phi_o.ref())
    res_ref = rhs_o.a - y.a
    out = lib.MultiFab(lay, lib.CELL, 1, 0)
    phi_d.fill_boundary(g_d)
    lib.abec_form(g_d, coef, 2, phi_d, rhs_d, out=out, **kw)
    got = out.gather_valid(n)
    assert np.array_equal(got, res_ref[..., 0] if res_ref.ndim == 4 and got.ndim == 3 else res_ref), float(np.abs(got - res_ref.reshape(got.shape)).max())
    # residual + restriction in one pass against residual, then cc_restrict
    nc = tuple(v // 2 for v in n)
    res_f = orc.Fab(n, orc.CELL, 0, 1); res_f.a[...] = res_ref
    crs_o = orc.Fab(nc, orc.CELL, 0, 1)
    L.orc_cc_restrict(crs_o.ref(), res_f.ref(), orc.i3(nc))
    clay = lib.Layout([(tuple(v // 2 for v in lo), tuple((v + 1) // 2 - 1 for v in hi)) for lo, hi in lay.boxes])
    crs_d = lib.MultiFab(clay, lib.CELL, 1, 0)
    lib.abec_form(g_d, coef, 3, phi_d, rhs_d, out=crs_d, **kw)
    got, ref = crs_d.gather_valid(nc), crs_o.valid(nc)
    assert np.array_equal(got, ref), ("restriction", float(np.abs(got - ref).max()))


@pytest.mark.parametrize("n,boxes", [((64, 64, 96), None), ((64, 64, 96), (32, 32, 48)), ((128, 64, 64), (64, 64, 32))])
def test_fused_tensor_residual_matches_the_oracle(orc, gpu, n, boxes):
    """constant viscosity: MLTensorOp::apply runs as ONE launch (k_tensor_cross_zm<.., FUSE>: 7-point part + cross terms from the velocity
    planes in LDS); a = 0 / b = -1 as in getTensorViscTerms (Diffusion.cpp:1655-1777) and the Crank-Nicolson form a = 1, b = dt"""
    lib = gpu
    from iamr_amd import ns as N
    L = orc.lib()
    g_o, g_d = orc.geom(n), lib.Geom.make(n)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    ax = [(np.arange(-1, n[d] + 1) + 0.5) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    rng = np.random.default_rng(4)
    u = orc.Fab(n, orc.CELL, 1, 3)
    u.a[..., 0] = np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) * np.cos(2 * np.pi * Z)
    u.a[..., 1] = np.cos(4 * np.pi * X) * np.sin(2 * np.pi * Y) + 0.3 * np.sin(2 * np.pi * Z)
    u.a[..., 2] = 0.5 * np.sin(2 * np.pi * (X + Y + Z))
    u.a[1:-1, 1:-1, 1:-1, :] += 0.01 * rng.standard_normal(tuple(n) + (3,))
    L.orc_fill_periodic(u.ref(), C.byref(g_o), orc.i3(orc.CELL))
    eta_o, eta_d = [], []
    for d in range(3):
        e = orc.Fab(n, orc.face(d), 0, 1, fill=0.013)
        eta_o.append(e)
        m = lib.MultiFab(lay, lib.face(d), 1, 0); m.setval(0.013); eta_d.append(m)
    acoef = orc.Fab(n, orc.CELL, 0, 1)
    acoef.a[..., 0] = 1.0 + 0.2 * np.cos(2 * np.pi * X[1:-1, 1:-1, 1:-1])
    a_d = lib.MultiFab(lay, lib.CELL, 1, 0); a_d.set_from_global(acoef.a, acoef.lo)
    u_d = lib.MultiFab(lay, lib.CELL, 3, 1); u_d.set_from_global(u.a, u.lo)
    for (a, b, ac_o, ac_d) in ((0.0, -1.0, None, None), (1.0, 0.004, acoef, a_d)):
        y = orc.Fab(n, orc.CELL, 0, 3)
        L.orc_tensor_apply(C.byref(g_o), y.ref(), u.ref(), C.c_double(a), C.c_double(b), ac_o.ref() if ac_o is not None else None, orc.fabptrs(eta_o))
        out_d = lib.MultiFab(lay, lib.CELL, 3, 0)
        N.tensor_apply(g_d, out_d, u_d, a, b, ac_d, eta_d)
        got = out_d.gather_valid(n)
        assert np.abs(got - y.a).max() <= 1e-12 * np.abs(y.a).max(), (a, b, float(np.abs(got - y.a).max()))


@pytest.mark.parametrize("coef", [1, 2])
@pytest.mark.parametrize("n", [(128, 32, 48), (256, 20, 16), (256, 64, 96)])
def test_one_launch_red_black_sweep_matches_the_oracle(orc, gpu, n, coef):
    """k_abec_gsrb_rb (round 4): a red and a black colour pass of an index-wrap level in one out-of-place launch -- rows of 128 / 256 cells in
    one / two wavefronts, new reds through registers, lane shuffles and an LDS ring, halo rows and the planes below / above every z-chunk
    recomputed; partial row tiles (20 rows in tiles of 6), several z-chunks.  The doubles of the oracle's two passes; also from a zero
    start without reading phi."""
    lib = gpu
    L = orc.lib()
    g_o, g_d = orc.geom(n), lib.Geom.make(n)
    lay = lib.Layout.single(n)
    rho, phi, rhs = fields(n, 9)
    scale, bu, beta = 0.37, (0.8, 1.1, 1.3), 1.0
    b_o = []
    for d in range(3):
        bf = orc.Fab(n, orc.face(d), 0, 1)
        if coef == 1:
            lo = [slice(1, n[e] + 1) for e in range(3)]; hi = [slice(1, n[e] + 1) for e in range(3)]
            lo[d] = slice(0, n[d] + 1); hi[d] = slice(1, n[d] + 2)
            bf.a[..., 0] = scale / (0.5 * (rho[tuple(lo)] + rho[tuple(hi)]))
        else:
            bf.a[...] = bu[d]
        b_o.append(bf)
    lev = orc.abec_level(g_o, b_o, beta=beta)
    rhs_o = orc.Fab(n, orc.CELL, 0, 1); rhs_o.a[..., 0] = rhs
    rho_d = lib.MultiFab(lay, lib.CELL, 1, 1); rho_d.set_from_global(rho[..., None], (-1,) * 3)
    rhs_d = lib.MultiFab(lay, lib.CELL, 1, 0); rhs_d.set_from_global(rhs[..., None], (0,) * 3)
    kw = dict(rho=rho_d, scale=scale, bu=bu, beta=beta)
    z3 = orc.i3([0, 0, 0])
    for start in ("field", "zero"):
        phi_o = orc.Fab(n, orc.CELL, 1, 1)
        phi_o.a[..., 0] = phi if start == "field" else 0.0
        a = lib.MultiFab(lay, lib.CELL, 1, 1); b = lib.MultiFab(lay, lib.CELL, 1, 1)
        a.set_from_global(phi[..., None], (-1,) * 3)          # zero start: the kernel must not read it
        b.setval(np.nan)
        for sweep in range(2):
            for rb in (0, 1):
                L.orc_fill_periodic(phi_o.ref(), C.byref(g_o), orc.i3(orc.CELL))
                L.orc_abec_gsrb(C.byref(lev), phi_o.ref(), rhs_o.ref(), rb, C.c_double(1.15), z3, z3, 3)
            lib.abec_form(g_d, coef, 7 if (start == "zero" and sweep == 0) else 6, a, rhs_d, out=b, **kw)
            a, b = b, a
            got, ref = a.gather_valid(n), phi_o.valid(n)
            assert np.array_equal(got, ref), (start, sweep, float(np.nanmax(np.abs(got - ref))), int(np.isnan(got).sum()))


PERIODIC, DIRICHLET, NEUMANN = 0, 101, 102
WALL_CASES = {
    # (periodic flags, low-side conditions, high-side conditions)
    "closed box (Neumann)": ((0, 0, 0), (NEUMANN,) * 3, (NEUMANN,) * 3),
    "walls in z": ((1, 1, 0), (PERIODIC, PERIODIC, NEUMANN), (PERIODIC, PERIODIC, NEUMANN)),
    "channel: inflow / outflow in x, walls in y": ((0, 0, 1), (NEUMANN, NEUMANN, PERIODIC), (DIRICHLET, NEUMANN, PERIODIC)),
    "Dirichlet everywhere": ((0, 0, 0), (DIRICHLET,) * 3, (DIRICHLET,) * 3),
}


@pytest.mark.parametrize("coef", [1, 2])
@pytest.mark.parametrize("case", list(WALL_CASES))
@pytest.mark.parametrize("n", [(128, 20, 16), (256, 32, 48)])
def test_one_launch_sweep_with_domain_walls_matches_the_oracle(orc, gpu, n, case, coef):
    """k_abec_gsrb_rb<., WALLS>: the one-launch sweep on a box that spans a domain with non-periodic sides.  It reads no ghost cell of phi
    (they are poisoned here): beyond a face it applies the ghost formula of the level's homogeneous boundary condition to the values at
    hand, as the ghost fill in front of each colour pass of the reference sequence does.  Against the oracle's fill + colour pass, twice."""
    lib = gpu
    L = orc.lib()
    per, lobc, hibc = WALL_CASES[case]
    g_o, g_d = orc.geom(n, periodic=per), lib.Geom.make(n, periodic=per)
    lay = lib.Layout.single(n)
    rho, phi, rhs = fields(n, 21)
    rng = np.random.default_rng(5)
    for d in range(3):                  # non-periodic sides: the density beyond the face is whatever the caller's fill left there
        if per[d]:
            continue
        for side in (0, -1):
            sl = [slice(None)] * 3; sl[d] = side
            rho[tuple(sl)] = 1.0 + 0.3 * rng.random(rho[tuple(sl)].shape)
    scale, bu, beta = 0.37, (0.8, 1.1, 1.3), 1.0
    b_o = []
    for d in range(3):
        bf = orc.Fab(n, orc.face(d), 0, 1)
        if coef == 1:
            lo = [slice(1, n[e] + 1) for e in range(3)]; hi = [slice(1, n[e] + 1) for e in range(3)]
            lo[d] = slice(0, n[d] + 1); hi[d] = slice(1, n[d] + 2)
            bf.a[..., 0] = scale / (0.5 * (rho[tuple(lo)] + rho[tuple(hi)]))
        else:
            bf.a[...] = bu[d]
        b_o.append(bf)
    lev = orc.abec_level(g_o, b_o, beta=beta)
    rhs_o = orc.Fab(n, orc.CELL, 0, 1); rhs_o.a[..., 0] = rhs
    rho_d = lib.MultiFab(lay, lib.CELL, 1, 1); rho_d.set_from_global(rho[..., None], (-1,) * 3)
    rhs_d = lib.MultiFab(lay, lib.CELL, 1, 0); rhs_d.set_from_global(rhs[..., None], (0,) * 3)
    kw = dict(rho=rho_d, scale=scale, bu=bu, beta=beta, lobc=lobc, hibc=hibc, maxorder=3)
    lo3, hi3 = orc.i3(lobc), orc.i3(hibc)
    for start in ("field", "zero"):
        phi_o = orc.Fab(n, orc.CELL, 1, 1)
        phi_o.a[..., 0] = phi if start == "field" else 0.0
        poisoned = phi.copy()
        for d in range(3):
            for side in (0, -1):
                sl = [slice(None)] * 3; sl[d] = side
                poisoned[tuple(sl)] = np.nan
        a = lib.MultiFab(lay, lib.CELL, 1, 1); b = lib.MultiFab(lay, lib.CELL, 1, 1)
        a.set_from_global(poisoned[..., None], (-1,) * 3)
        b.setval(np.nan)
        for sweep in range(2):
            for rb in (0, 1):
                L.orc_fill_periodic(phi_o.ref(), C.byref(g_o), orc.i3(orc.CELL))
                L.orc_abec_applybc(C.byref(lev), phi_o.ref(), lo3, hi3, 3, 0, None)
                L.orc_abec_gsrb(C.byref(lev), phi_o.ref(), rhs_o.ref(), rb, C.c_double(1.15), lo3, hi3, 3)
            lib.abec_form(g_d, coef, 7 if (start == "zero" and sweep == 0) else 6, a, rhs_d, out=b, **kw)
            a, b = b, a
            got, ref = a.gather_valid(n), phi_o.valid(n)
            assert np.array_equal(got, ref), (case, start, sweep, float(np.nanmax(np.abs(got - ref))), int(np.isnan(got).sum()))


@pytest.mark.parametrize("coef", [1, 2])
@pytest.mark.parametrize("maxorder", [2, 3, 4])
@pytest.mark.parametrize("nb", [(128, 20, 16), (256, 32, 48)])
def test_one_launch_sweep_on_a_refined_box_matches_the_oracle(orc, gpu, nb, maxorder, coef):
    """k_abec_gsrb_rb<., WALLS, ., ., W3> (round 5): the one-launch sweep on a refined level that is one box strictly inside its domain --
    every face a coarse/fine face.  The kernel reads no ghost cell of phi (poisoned here): beyond a face it evaluates the homogeneous
    coarse/fine ghost value of the level's maxorder (up to three cells deep: face cell, the cell behind it, the one behind that) on the
    values at hand, as the fill in front of each colour pass of the reference sequence does (MLLinOp::setCoarseFineBC on a level set
    up as in MacProj.cpp:1166-1170; homogeneous inside MLMG::mgVcycle).  Against the oracle's colour passes with its on-the-fly
    coarse/fine ghost values (orc_abec_gsrb on a level with a box list), two sweeps, from a field and from a zero start."""
    lib = gpu
    L = orc.lib()
    off = 8
    n = tuple(v + 2 * off for v in nb)
    box = (tuple([off] * 3), tuple(off + v - 1 for v in nb))
    g_o, g_d = orc.geom(n), lib.Geom.make(n)
    lay = lib.Layout([box])
    rho, phi, rhs = fields(n, 33 + maxorder)
    scale, bu, beta = 0.37, (0.8, 1.1, 1.3), 1.0
    b_o = []
    for d in range(3):
        bf = orc.Fab(n, orc.face(d), 0, 1)
        if coef == 1:
            lo = [slice(1, n[e] + 1) for e in range(3)]; hi = [slice(1, n[e] + 1) for e in range(3)]
            lo[d] = slice(0, n[d] + 1); hi[d] = slice(1, n[d] + 2)
            bf.a[..., 0] = scale / (0.5 * (rho[tuple(lo)] + rho[tuple(hi)]))
        else:
            bf.a[...] = bu[d]
        b_o.append(bf)
    lev = orc.abec_level(g_o, b_o, beta=beta, boxes=[box], ratio=2)
    rhs_o = orc.Fab(n, orc.CELL, 0, 1); rhs_o.a[..., 0] = rhs
    rho_d = lib.MultiFab(lay, lib.CELL, 1, 1); rho_d.set_from_global(rho[..., None], (-1,) * 3)
    rhs_d = lib.MultiFab(lay, lib.CELL, 1, 0); rhs_d.set_from_global(rhs[..., None], (0,) * 3)
    kw = dict(rho=rho_d, scale=scale, bu=bu, beta=beta, maxorder=maxorder)
    z3 = orc.i3([0, 0, 0])
    inside = tuple(slice(off, off + v) for v in nb)
    for start in ("field", "zero"):
        phi_o = orc.Fab(n, orc.CELL, 1, 1)
        phi_o.a[..., 0] = phi if start == "field" else 0.0
        a = lib.MultiFab(lay, lib.CELL, 1, 1); b = lib.MultiFab(lay, lib.CELL, 1, 1)
        loc = np.full(tuple(v + 2 for v in nb) + (1,), np.nan)          # the box with its ghost layer: ghost cells poisoned
        loc[1:-1, 1:-1, 1:-1, 0] = phi[1:-1, 1:-1, 1:-1][inside]
        a.from_numpy(loc)                                                # (zero start: the kernel must not read it at all)
        b.setval(np.nan)
        for sweep in range(2):
            for rb in (0, 1):
                L.orc_abec_applybc(C.byref(lev), phi_o.ref(), z3, z3, maxorder, 0, None)      # (sets the homogeneous form and the order)
                L.orc_abec_gsrb(C.byref(lev), phi_o.ref(), rhs_o.ref(), rb, C.c_double(1.15), z3, z3, maxorder)
            lib.abec_form(g_d, coef, 11 if (start == "zero" and sweep == 0) else 10, a, rhs_d, out=b, **kw)
            a, b = b, a
            got, ref = a.gather_valid(n)[..., 0][inside], phi_o.valid(n)[..., 0][inside]
            assert not np.isnan(got).any(), (start, sweep, int(np.isnan(got).sum()))
            assert np.array_equal(got, ref), (maxorder, start, sweep, float(np.abs(got - ref).max()), int((got != ref).sum()))
