"""GPU parity: cell-centred ABec kernels, MLMG solve and MAC projection (HIP, through the C-ABI)
against the CPU oracle on identical seeded inputs.  Tolerances are stated per test."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def smooth_field(n, ng, seed, typ=(0, 0, 0)):
    """deterministic periodic smooth + rough field on the grown index region"""
    rng = np.random.default_rng(seed)
    ax = [(np.arange(-ng, n[d] + typ[d] + ng) + (0.0 if typ[d] else 0.5)) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    f = np.sin(2 * np.pi * X) * np.cos(4 * np.pi * Y) + 0.5 * np.cos(2 * np.pi * (Z + X)) + 0.3 * np.sin(6 * np.pi * Y * 1.0)
    return f + 0.1 * rng.standard_normal(f.shape)


def make_problem(orc, lib, n, seed=0, layout=None, varb=True):
    """variable-coefficient periodic problem on both sides; returns dicts of oracle fabs and device multifabs"""
    g_o = orc.geom(n)
    g_d = lib.Geom.make(n)
    lay = layout if layout is not None else lib.Layout.single(n)
    rng = np.random.default_rng(seed)
    o = {}
    d = {}
    # b coefficients on faces (positive), periodic-consistent
    o["b"], d["b"] = [], []
    for dd in range(3):
        t = orc.face(dd)
        bf = orc.Fab(n, t, 0, 1)
        ax = [(np.arange(0, n[q] + t[q]) + (0.0 if t[q] else 0.5)) / n[q] for q in range(3)]
        X, Y, Z = np.meshgrid(*ax, indexing="ij")
        vals = 1.0 + (0.5 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) * np.cos(4 * np.pi * Z) if varb else 0.0)
        bf.a[..., 0] = vals
        o["b"].append(bf)
        m = lib.MultiFab(lay, t, 1, 0)
        m.set_from_global(bf.a, bf.lo)
        d["b"].append(m)
    phi = orc.Fab(n, orc.CELL, 1, 1)
    phi.a[..., 0] = smooth_field(n, 1, seed + 1)
    rhs = orc.Fab(n, orc.CELL, 0, 1)
    rhs.a[..., 0] = smooth_field(n, 0, seed + 2)
    o["phi"], o["rhs"] = phi, rhs
    d["phi"] = lib.MultiFab(lay, lib.CELL, 1, 1)
    d["phi"].set_from_global(phi.a, phi.lo)
    d["rhs"] = lib.MultiFab(lay, lib.CELL, 1, 0)
    d["rhs"].set_from_global(rhs.a, rhs.lo)
    return g_o, g_d, lay, o, d


@pytest.mark.parametrize("n,boxes", [((32, 32, 32), None), ((32, 16, 24), None), ((32, 32, 32), 16)])
def test_gsrb_and_residual_match_oracle(orc, gpu, n, boxes):
    lib = gpu
    L = orc.lib()
    lay = lib.Layout.decompose(n, boxes) if boxes else None
    g_o, g_d, lay, o, d = make_problem(orc, lib, n, seed=3, layout=lay)
    lev = orc.abec_level(g_o, o["b"])
    z3 = orc.i3([0, 0, 0])
    # ghost fill on both sides, then red and black passes
    for rb in (0, 1):
        L.orc_fill_periodic(o["phi"].ref(), C.byref(g_o), orc.i3(orc.CELL))
        L.orc_abec_gsrb(C.byref(lev), o["phi"].ref(), o["rhs"].ref(), rb, C.c_double(1.15), z3, z3, 3)
        d["phi"].fill_boundary(g_d)
        lib.abec_gsrb(g_d, 0.0, 1.0, None, d["b"], d["phi"], d["rhs"], rb, 1.15)
    got = d["phi"].gather_valid(n)
    ref = o["phi"].valid(n)
    # bit-exact: same expression order, -ffp-contract=off on both sides
    assert np.array_equal(got, ref), np.abs(got - ref).max()
    # residual
    L.orc_fill_periodic(o["phi"].ref(), C.byref(g_o), orc.i3(orc.CELL))
    y = orc.Fab(n, orc.CELL, 0, 1)
    L.orc_abec_apply(C.byref(lev), y.ref(), o["phi"].ref())
    ref_res = o["rhs"].a - y.a
    out = lib.MultiFab(lay, lib.CELL, 1, 0)
    d["phi"].fill_boundary(g_d)
    lib.abec_residual(g_d, 0.0, 1.0, None, d["b"], out, d["phi"], d["rhs"])
    got_res = out.gather_valid(n)
    assert np.array_equal(got_res, ref_res), np.abs(got_res - ref_res).max()


def test_mlmg_solve_matches_oracle(orc, gpu):
    lib = gpu
    L = orc.lib()
    n = (32, 32, 32)
    g_o, g_d, lay, o, d = make_problem(orc, lib, n, seed=7)
    o["phi"].a[...] = 0.0
    d["phi"].setval(0.0)
    lev = orc.abec_level(g_o, o["b"])
    z3 = orc.i3([0, 0, 0])
    st_o = orc.CMgStats()
    oo = orc.mg_opts()
    L.orc_abec_solve(C.byref(lev), o["phi"].ref(), o["rhs"].ref(), z3, z3, C.c_double(1e-12), C.c_double(1e-16), C.byref(oo), C.byref(st_o))
    st = lib.abec_solve(g_d, 0.0, 1.0, None, d["b"], d["phi"], d["rhs"], rtol=1e-12, atol=1e-16)
    assert st.converged == 1 and st_o.converged == 1
    assert st.iters == st_o.iters
    got = d["phi"].gather_valid(n)[..., 0]
    ref = o["phi"].valid(n)[..., 0]
    got = got - got.mean()
    ref = ref - ref.mean()
    # both solve to rtol 1e-12; the bottom BiCGStab reductions differ in summation order
    assert np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max()
    # discrete invariant: residual reduced below tolerance
    assert st.resnorm <= 1e-12 * max(st.rhsnorm0, st.resnorm0) * 1.0000001


@pytest.mark.parametrize("boxes", [None, 16])
def test_mac_project_matches_oracle(orc, gpu, boxes):
    lib = gpu
    L = orc.lib()
    n = (32, 32, 32)
    g_o = orc.geom(n)
    g_d = lib.Geom.make(n)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    um_o, um_d = [], []
    for dd in range(3):
        t = orc.face(dd)
        f = orc.Fab(n, t, 1, 1)
        f.a[..., 0] = smooth_field(n, 1, 11 + dd, t)
        L.orc_fill_periodic(f.ref(), C.byref(g_o), orc.i3(t))
        # make the periodic duplicate face consistent
        sl_hi = [slice(None)] * 3
        sl_lo = [slice(None)] * 3
        sl_hi[dd] = 1 + n[dd]
        sl_lo[dd] = 1
        f.a[tuple(sl_hi)] = f.a[tuple(sl_lo)]
        L.orc_fill_periodic(f.ref(), C.byref(g_o), orc.i3(t))
        um_o.append(f)
        m = lib.MultiFab(lay, t, 1, 1)
        m.set_from_global(f.a, f.lo)
        um_d.append(m)
    rho = orc.Fab(n, orc.CELL, 1, 1)
    rho.a[..., 0] = 1.0 + 0.3 * np.sin(2 * np.pi * smooth_field(n, 1, 5) * 0.1)
    L.orc_fill_periodic(rho.ref(), C.byref(g_o), orc.i3(orc.CELL))
    rho_d = lib.MultiFab(lay, lib.CELL, 1, 1)
    rho_d.set_from_global(rho.a, rho.lo)
    phi_o = orc.Fab(n, orc.CELL, 1, 1)
    phi_d = lib.MultiFab(lay, lib.CELL, 1, 1)
    phi_d.setval(0.0)
    dt = 0.01
    z3 = orc.i3([0, 0, 0])
    st_o = orc.CMgStats()
    oo = orc.mg_opts(maxorder=4)
    L.orc_mac_project(C.byref(g_o), orc.fabptrs(um_o), rho.ref(), None, phi_o.ref(), C.c_double(2.0 / dt), z3, z3,
                      C.c_double(1e-12), C.c_double(1e-16), C.byref(oo), C.byref(st_o))
    st = lib.mlmg_mac_solve(g_d, um_d, rho_d, 0, None, phi_d, 2.0 / dt)
    assert st.converged == 1
    # invariant (MacProj::check_div_cond): projected field is discretely divergence free
    div = lib.MultiFab(lay, lib.CELL, 1, 0)
    lib.mac_divergence(g_d, div, um_d)
    assert div.norm0() <= 1e-10 * max(st.rhsnorm0, 1e-30) + 1e-12
    for dd in range(3):
        got = um_d[dd].gather_valid(n)[..., 0]
        ref = um_o[dd].valid(n, orc.face(dd))[..., 0]
        assert np.abs(got - ref).max() <= 1e-9 * np.abs(ref).max()
