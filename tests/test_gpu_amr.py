"""GPU: inter-level building blocks of the AMR hierarchy: ParallelCopy between different box layouts (with periodic
images) and average_down of cell / face / nodal data from a fine patch onto the coarse level, against direct numpy evaluation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_parallel_copy_between_layouts_with_periodic_images(gpu):
    lib = gpu
    n = (16, 12, 8)
    g = lib.Geom.make(n)
    src_lay = lib.Layout.decompose(n, (8, 6, 8))          # 4 boxes
    dst_lay = lib.Layout([((0, 0, 0), (15, 3, 7)), ((0, 4, 0), (15, 11, 7))], [0, 0])
    rng = np.random.default_rng(2)
    G = rng.standard_normal(n + (3,))
    src = lib.MultiFab(src_lay, lib.CELL, 3, 0)
    src.set_from_global(G, (0, 0, 0))
    dst = lib.MultiFab(dst_lay, lib.CELL, 2, 2)
    dst.setval(-7.0)
    lib.parallel_copy(dst, src, scomp=1, dcomp=0, ncomp=2, src_ng=0, dst_ng=2, periodic_geom=g)
    for li in range(dst.nlocal()):
        a, lo = dst.to_numpy(li)
        idx = [np.mod(np.arange(lo[d], lo[d] + a.shape[d]), n[d]) for d in range(3)]
        assert np.array_equal(a, G[np.ix_(*idx)][..., 1:3])
    # without periodic images only the part inside the domain is touched
    dst.setval(-7.0)
    lib.parallel_copy(dst, src, scomp=0, dcomp=1, ncomp=1, src_ng=0, dst_ng=2)
    a, lo = dst.to_numpy(0)
    assert np.all(a[:2] == -7.0) and np.all(a[..., 0] == -7.0)
    assert np.array_equal(a[2:-2, 2:, 2:-2, 1], G[:, 0:6, :, 0])


@pytest.mark.parametrize("typ", ["cell", "facex", "facez", "node"])
def test_average_down_fine_patch_onto_coarse_level(gpu, typ):
    """NavierStokesBase::avgDown_StatePress pieces: state / Gradp (cells), pressure (nodes, injection), face data; ratio 2;
    the fine patch is chopped differently from the coarse level"""
    lib = gpu
    nc = (16, 16, 16)
    crse_lay = lib.Layout.decompose(nc, (8, 16, 16))
    # fine patch = coarse cells [4..11] x [2..9] x [0..15]  ->  fine cells [8..23] x [4..19] x [0..31], 4 fine boxes
    fine_boxes = [((8, 4, 0), (15, 19, 15)), ((16, 4, 0), (23, 19, 15)), ((8, 4, 16), (15, 19, 31)), ((16, 4, 16), (23, 19, 31))]
    fine_lay = lib.Layout(fine_boxes, [0] * 4)
    t = {"cell": lib.CELL, "facex": lib.face(0), "facez": lib.face(2), "node": lib.NODE}[typ]
    ncomp = 3
    rng = np.random.default_rng(4)
    nf = tuple(2 * nc[d] + t[d] for d in range(3))
    F = rng.standard_normal(nf + (ncomp,))
    ncs = tuple(nc[d] + t[d] for d in range(3))
    Cg = rng.standard_normal(ncs + (ncomp,))
    fine = lib.MultiFab(fine_lay, t, ncomp, 0); fine.set_from_global(F, (0, 0, 0))
    crse = lib.MultiFab(crse_lay, t, ncomp, 0); crse.set_from_global(Cg, (0, 0, 0))
    lib.average_down(fine, crse, scomp=1, ncomp=2, ratio=2)
    exp = Cg.copy()
    # coarse index ranges covered by the patch, per index type
    clo, chi = (4, 2, 0), (11, 9, 15)
    for i in range(clo[0], chi[0] + 1 + t[0]):
        for j in range(clo[1], chi[1] + 1 + t[1]):
            for k in range(clo[2], chi[2] + 1 + t[2]):
                r = [range(2 * c, 2 * c + (1 if t[d] else 2)) for d, c in enumerate((i, j, k))]
                exp[i, j, k, 1:3] = F[np.ix_(*r)][..., 1:3].reshape(-1, 2).mean(axis=0)
    got = crse.gather_valid(nc)
    assert np.array_equal(got[..., 0], Cg[..., 0])                       # untouched component
    assert np.abs(got[..., 1:3] - exp[..., 1:3]).max() <= 1e-15
    assert np.abs(got - Cg).max() > 0.1


def test_fillpatch_two_levels_matches_oracle(orc, gpu):
    """AmrLevel::FillPatch on a refined level: same-level copy, periodic image, conservative-linear (linear-limited) interpolation
    from the time-interpolated coarse level, physical BC -- all ghost cells of a 2-box fine patch that touches a periodic
    boundary (x-lo) and a wall (y-lo) against the CPU oracle.  Tolerance: 1e-13 (same formulas, different summation order of
    the time interpolation)."""
    import ctypes as C
    lib = gpu
    L = orc.lib()
    nc, nf, ncomp, ng, ratio = (16, 16, 16), (32, 32, 32), 3, 3, 2
    per = (1, 0, 1)
    cg = lib.Geom.make(nc, periodic=per)
    fg = lib.Geom.make(nf, periodic=per)
    crse_lay = lib.Layout.decompose(nc, (8, 16, 8))
    # fine patch: coarse cells [0..7] x [0..7] x [4..11]  ->  fine [0..15] x [0..15] x [8..23], two boxes split in z
    fboxes = [((0, 0, 8), (15, 15, 15)), ((0, 0, 16), (15, 15, 23))]
    fine_lay = lib.Layout(fboxes, [0, 0])
    # comp 0: foextrap at the walls, comp 1: ext_dir (values 0.3 / -0.2), comp 2: hoextrap
    bcs = [((0, 2, 0), (0, 2, 0)), ((0, 3, 0), (0, 3, 0)), ((0, 4, 0), (0, 4, 0))]
    edlo = [[0.0, 0.0, 0.0], [0.0, 0.3, 0.0], [0.0, 0.0, 0.0]]
    edhi = [[0.0, 0.0, 0.0], [0.0, -0.2, 0.0], [0.0, 0.0, 0.0]]
    rng = np.random.default_rng(8)
    xc = [(np.arange(nc[d]) + 0.5) / nc[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*xc, indexing="ij")
    def smooth(seed):
        r = np.random.default_rng(seed)
        out = np.zeros(nc + (ncomp,))
        for n in range(ncomp):
            a, b, c = r.random(3)
            out[..., n] = np.sin(2 * np.pi * (X + a)) * np.cos(np.pi * (Y + b)) * np.sin(2 * np.pi * (Z + c)) + 0.3 * r.standard_normal(nc)
        return out
    Cold, Cnew = smooth(1), smooth(2)
    Fnew = rng.standard_normal(nf + (ncomp,))
    t_co, t_cn, time = 0.0, 1.0, 0.3
    cold = lib.MultiFab(crse_lay, lib.CELL, ncomp, 0); cold.set_from_global(Cold, (0, 0, 0))
    cnew = lib.MultiFab(crse_lay, lib.CELL, ncomp, 0); cnew.set_from_global(Cnew, (0, 0, 0))
    fnew = lib.MultiFab(fine_lay, lib.CELL, ncomp, 0); fnew.set_from_global(Fnew, (0, 0, 0))
    dst = lib.MultiFab(fine_lay, lib.CELL, ncomp, ng); dst.setval(-99.0)
    lib.fillpatch_two_levels(dst, time, (None, fnew, time, time), (cold, cnew, t_co, t_cn), cg, fg, bc=bcs, extdir_lo=edlo, extdir_hi=edhi, ratio=ratio)
    # ---- oracle
    g_c = orc.geom(nc, periodic=per)
    g_f = orc.geom(nf, periodic=per)
    cf = orc.Fab(nc, orc.CELL, 4, ncomp)
    cf.a[4:-4, 4:-4, 4:-4, :] = (t_cn - time) / (t_cn - t_co) * Cold + (time - t_co) / (t_cn - t_co) * Cnew
    L.orc_fill_periodic(cf.ref(), C.byref(g_c), orc.i3(orc.CELL))
    bcr = (orc.CBCRec * ncomp)()
    for n in range(ncomp):
        bcr[n].lo = (C.c_int * 3)(*bcs[n][0]); bcr[n].hi = (C.c_int * 3)(*bcs[n][1])
    el = (C.c_double * 9)(*[v for row in edlo for v in row]); eh = (C.c_double * 9)(*[v for row in edhi for v in row])
    L.orc_fill_physbc_cc(cf.ref(), C.byref(g_c), bcr, el, eh)
    vlo, vhi = (0, 0, 8), (15, 15, 23)
    ff = orc.Fab(nf, orc.CELL, 0, ncomp, lo=tuple(v - ng for v in vlo), hi=tuple(v + ng for v in vhi))
    ff.a[...] = -99.0
    ff.a[ng:-ng, ng:-ng, ng:-ng, :] = Fnew[vlo[0]:vhi[0] + 1, vlo[1]:vhi[1] + 1, vlo[2]:vhi[2] + 1]
    flo = [vlo[d] - ng if per[d] else max(vlo[d] - ng, 0) for d in range(3)]
    fhi = [vhi[d] + ng if per[d] else min(vhi[d] + ng, nf[d] - 1) for d in range(3)]
    # periodic image of the fine patch itself: x spans [0,15] only, so ghost cells at x < 0 come from the coarse level
    L.orc_fill_coarse_fine(ff.ref(), orc.i3(flo), orc.i3(fhi), orc.i3(vlo), orc.i3(vhi), cf.ref(), orc.i3((0, 0, 0)), orc.i3(tuple(v - 1 for v in nc)),
                           orc.i3(per), ratio, bcr)
    L.orc_fill_physbc_cc(ff.ref(), C.byref(g_f), bcr, el, eh)
    # ---- compare every cell (valid + ghost) of both fine boxes
    for li in range(dst.nlocal()):
        a, lo = dst.to_numpy(li)
        sl = tuple(slice(lo[d] - ff.lo[d], lo[d] - ff.lo[d] + a.shape[d]) for d in range(3))
        ref = ff.a[sl]
        assert np.abs(ref).max() < 50.0                        # every cell was filled by the oracle
        assert np.abs(a - ref).max() <= 1e-13, (li, np.abs(a - ref).max())


def test_flux_register_reflux_matches_direct_evaluation(gpu):
    """FluxRegister CrseInit(-dt_c) + two FineAdd(+dt_f) + Reflux, for a two-box fine patch that touches a periodic boundary;
    expected coarse-cell corrections evaluated directly with numpy.  Also the telescoping property: with fine fluxes that are
    the exact refinement of the coarse ones the registers cancel and Reflux changes nothing."""
    lib = gpu
    nc, ratio, ncomp = (16, 12, 8), 2, 2
    cg = lib.Geom.make(nc)
    crse_lay = lib.Layout.decompose(nc, (8, 12, 8))
    # coarse cells covered by the patch: [0..7] x [2..9] x [2..5]  (x-lo face on the periodic boundary)
    cboxes = [((0, 2, 2), (3, 9, 5)), ((4, 2, 2), (7, 9, 5))]
    fboxes = [(tuple(2 * v for v in lo), tuple(2 * v + 1 for v in hi)) for lo, hi in cboxes]
    fine_lay = lib.Layout(fboxes, [0, 0])
    rng = np.random.default_rng(6)
    nf = tuple(2 * v for v in nc)
    dt_c, dt_f, vol = 0.1, 0.05, 0.25
    CF, FF1, FF2, cf_mf, ff1_mf, ff2_mf = [], [], [], [], [], []
    for d in range(3):
        sc = list(nc); sc[d] += 1
        sf = list(nf); sf[d] += 1
        c = rng.standard_normal(tuple(sc) + (ncomp,))
        # periodic consistency of the coarse flux on the domain faces
        lo = [slice(None)] * 3; hi = [slice(None)] * 3
        lo[d] = 0; hi[d] = nc[d]
        c[tuple(hi)] = c[tuple(lo)]
        CF.append(c); FF1.append(rng.standard_normal(tuple(sf) + (ncomp,))); FF2.append(rng.standard_normal(tuple(sf) + (ncomp,)))
        m = lib.MultiFab(crse_lay, lib.face(d), ncomp, 0); m.set_from_global(c, (0, 0, 0)); cf_mf.append(m)
        m = lib.MultiFab(fine_lay, lib.face(d), ncomp, 0); m.set_from_global(FF1[d], (0, 0, 0)); ff1_mf.append(m)
        m = lib.MultiFab(fine_lay, lib.face(d), ncomp, 0); m.set_from_global(FF2[d], (0, 0, 0)); ff2_mf.append(m)
    fr = lib.FluxRegister(fine_lay, crse_lay, cg, ratio, ncomp)
    for d in range(3):
        fr.CrseInit(cf_mf[d], d, 0, 0, ncomp, -dt_c)
        fr.FineAdd(ff1_mf[d], d, 0, 0, ncomp, dt_f)
        fr.FineAdd(ff2_mf[d], d, 0, 0, ncomp, dt_f)
    S = lib.MultiFab(crse_lay, lib.CELL, ncomp, 0); S.setval(0.0)
    fr.Reflux(S, vol, 1.0, 0, 0, ncomp)
    exp = np.zeros(nc + (ncomp,))
    for (lo, hi) in cboxes:
        for d in range(3):
            d1, d2 = [e for e in range(3) if e != d]
            for side in (0, 1):
                face = lo[d] if side == 0 else hi[d] + 1
                out = (lo[d] - 1) % nc[d] if side == 0 else (hi[d] + 1) % nc[d]
                for a in range(lo[d1], hi[d1] + 1):
                    for b in range(lo[d2], hi[d2] + 1):
                        ci = [0, 0, 0]; ci[d] = face; ci[d1] = a; ci[d2] = b
                        fsum = np.zeros(ncomp)
                        for ra in range(2):
                            for rb in range(2):
                                fi = [0, 0, 0]; fi[d] = 2 * face; fi[d1] = 2 * a + ra; fi[d2] = 2 * b + rb
                                fsum += FF1[d][tuple(fi)] + FF2[d][tuple(fi)]
                        reg = -dt_c * CF[d][tuple(ci)] + dt_f * fsum
                        oc = [0, 0, 0]; oc[d] = out; oc[d1] = a; oc[d2] = b
                        exp[tuple(oc)] += (-1.0 if side == 0 else 1.0) * reg / vol
    got = S.gather_valid(nc)
    assert np.abs(exp).max() > 0.5
    assert np.abs(got - exp).max() <= 1e-13
    # telescoping: fine flux = coarse flux / 4 on every child face, dt_f = dt_c / 2, two fine steps
    fr.setVal(0.0)
    for d in range(3):
        F = np.repeat(np.repeat(np.repeat(CF[d], 2, axis=0), 2, axis=1), 2, axis=2)
        sl = [slice(None)] * 3
        sl[d] = slice(0, 2 * nc[d] + 1, 1)
        # refine the face-normal direction by injection: fine face 2i <-> coarse face i
        idx = [np.arange(nf[e] + (1 if e == d else 0)) // 2 for e in range(3)]
        idx[d] = np.minimum(np.arange(nf[d] + 1) // 2 + (np.arange(nf[d] + 1) % 2), nc[d])   # odd fine faces are never used by the register
        Fd = CF[d][np.ix_(*idx)] * 0.25
        m = lib.MultiFab(fine_lay, lib.face(d), ncomp, 0); m.set_from_global(Fd, (0, 0, 0))
        fr.CrseInit(cf_mf[d], d, 0, 0, ncomp, -dt_c)
        fr.FineAdd(m, d, 0, 0, ncomp, dt_f)
        fr.FineAdd(m, d, 0, 0, ncomp, dt_f)
    S.setval(0.0)
    fr.Reflux(S, vol, 1.0, 0, 0, ncomp)
    assert np.abs(S.gather_valid(nc)).max() <= 1e-14


def test_two_level_subcycled_advection_conserves_mass(gpu):
    """The a18 building blocks working together on IAMR's subcycling pattern for a conservatively advected scalar (density):
    coarse step with CrseInit(-dt), two fine substeps whose ghost cells come from FillPatchTwoLevels (time-interpolated coarse
    data, conservative-linear interpolation) with FineAdd(+dt/2), then Reflux and average_down (NavierStokes::reflux / avgDown,
    reference Source/NavierStokes.cpp:1736-1873).  With a divergence-free (uniform) mac velocity on a periodic domain the total
    mass of the composite grid is conserved to round-off -- without the Reflux step it is not."""
    lib = gpu
    nc, ratio = (16, 16, 16), 2
    nf = tuple(ratio * v for v in nc)
    cg = lib.Geom.make(nc)
    fg = lib.Geom.make(nf)
    crse_lay = lib.Layout.decompose(nc, (8, 16, 16))
    cboxes = [((4, 4, 4), (7, 11, 11)), ((8, 4, 4), (11, 11, 11))]
    fboxes = [(tuple(2 * v for v in lo), tuple(2 * v + 1 for v in hi)) for lo, hi in cboxes]
    fine_lay = lib.Layout(fboxes, [0, 0])
    u = (0.7, -0.4, 0.3)
    dxc = 1.0 / nc[0]
    dt_c = 0.4 * dxc / max(abs(v) for v in u)
    dt_f = dt_c / ratio
    volc, volf = dxc ** 3, (dxc / ratio) ** 3

    def make_umac(lay):
        um = []
        for d in range(3):
            m = lib.MultiFab(lay, lib.face(d), 1, 1); m.setval(u[d]); um.append(m)
        return um
    umc, umf = make_umac(crse_lay), make_umac(fine_lay)
    xc = [(np.arange(nc[d]) + 0.5) / nc[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*xc, indexing="ij")
    rho_c = 1.0 + 0.5 * np.exp(-60.0 * ((X - 0.45) ** 2 + (Y - 0.5) ** 2 + (Z - 0.5) ** 2))
    xf = [(np.arange(nf[d]) + 0.5) / nf[d] for d in range(3)]
    Xf, Yf, Zf = np.meshgrid(*xf, indexing="ij")
    rho_f = 1.0 + 0.5 * np.exp(-60.0 * ((Xf - 0.45) ** 2 + (Yf - 0.5) ** 2 + (Zf - 0.5) ** 2))
    Sc_old = lib.MultiFab(crse_lay, lib.CELL, 1, 0); Sc_old.set_from_global(rho_c[..., None], (0, 0, 0))
    Sf = lib.MultiFab(fine_lay, lib.CELL, 1, 0); Sf.set_from_global(rho_f[..., None], (0, 0, 0))
    lib.average_down(Sf, Sc_old, 0, 1, ratio)               # consistent composite initial data

    def covered_mask():
        m = np.zeros(nc, bool)
        for lo, hi in cboxes:
            m[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1] = True
        return m
    cov = covered_mask()

    def composite_mass(Sc, Sfine):
        c = Sc.gather_valid(nc)[..., 0]
        tot = c[~cov].sum() * volc
        for li in range(Sfine.nlocal()):
            a, lo = Sfine.to_numpy(li)
            tot += a.sum() * volf
        return tot
    m0 = composite_mass(Sc_old, Sf)

    def advect(geom, lay, S_valid_src, fill, um, dt):
        """one conservative Godunov update; returns (new S, fluxes)"""
        Sg = lib.MultiFab(lay, lib.CELL, 1, 3)
        fill(Sg)
        aofs = lib.MultiFab(lay, lib.CELL, 1, 0)
        flux = [lib.MultiFab(lay, lib.face(d), 1, 0) for d in range(3)]
        lib.godunov_compute_aofs(geom, aofs, 0, Sg, 1, None, None, um, (1,), dt, None, 0, 0, None, flux)
        Snew = lib.MultiFab(lay, lib.CELL, 1, 0)
        for li in range(Snew.nlocal()):
            a, lo = S_valid_src.to_numpy(li)
            b, _ = aofs.to_numpy(li)
            Snew.from_numpy(a - dt * b, li)
        return Snew, flux

    def run(do_reflux):
        fr = lib.FluxRegister(fine_lay, crse_lay, cg, ratio, 1)
        # coarse advance
        def fill_c(Sg):
            lib.parallel_copy(Sg, Sc_old, 0, 0, 1, 0, 3, cg)
        Sc_new, cflux = advect(cg, crse_lay, Sc_old, fill_c, umc, dt_c)
        for d in range(3):
            fr.CrseInit(cflux[d], d, 0, 0, 1, -dt_c)
        # fine substeps
        Sf_cur = Sf
        for it in range(ratio):
            t = it * dt_f
            def fill_f(Sg, Sf_cur=Sf_cur, t=t):
                lib.fillpatch_two_levels(Sg, t, (None, Sf_cur, t, t), (Sc_old, Sc_new, 0.0, dt_c), cg, fg, ratio=ratio)
            Sf_cur, fflux = advect(fg, fine_lay, Sf_cur, fill_f, umf, dt_f)
            for d in range(3):
                fr.FineAdd(fflux[d], d, 0, 0, 1, dt_f)
        if do_reflux:
            fr.Reflux(Sc_new, volc, 1.0, 0, 0, 1)
        lib.average_down(Sf_cur, Sc_new, 0, 1, ratio)
        return composite_mass(Sc_new, Sf_cur), Sc_new
    m1, Sc1 = run(True)
    m_noreflux, _ = run(False)
    assert abs(m1 - m0) <= 1e-13 * m0, (m1 - m0)
    assert abs(m_noreflux - m0) > 1e-8 * m0            # the registers matter
    assert np.abs(Sc1.gather_valid(nc)[..., 0] - rho_c).max() > 1e-3      # something was advected


def test_create_umac_grown_on_refined_level_matches_oracle(orc, gpu):
    """NavierStokesBase::create_umac_grown for level > 0: FaceLinear coarse-fine fill of the ghost faces of a two-box fine patch and
    IAMR's in-tree divergence fix; afterwards every face-adjacent ghost cell is divergence free.  Bit-exact vs the oracle
    (evaluated on the union of the two boxes)."""
    import ctypes as C
    lib = gpu
    L = orc.lib()
    nc, ratio = (16, 16, 16), 2
    nf = tuple(ratio * v for v in nc)
    cg = lib.Geom.make(nc); fg = lib.Geom.make(nf)
    crse_lay = lib.Layout.decompose(nc, (8, 16, 16))
    vlo, vhi = (8, 12, 8), (23, 19, 23)                          # fine valid region = union of the two boxes
    fboxes = [((8, 12, 8), (15, 19, 23)), ((16, 12, 8), (23, 19, 23))]
    fine_lay = lib.Layout(fboxes, [0, 0])
    rng = np.random.default_rng(12)
    UC, UF, uc_mf, uf_mf = [], [], [], []
    for d in range(3):
        sc = list(nc); sc[d] += 1
        c = rng.standard_normal(tuple(sc))
        lo = [slice(None)] * 3; hi = [slice(None)] * 3
        lo[d] = 0; hi[d] = nc[d]
        c[tuple(hi)] = c[tuple(lo)]
        UC.append(c)
        m = lib.MultiFab(crse_lay, lib.face(d), 1, 0); m.set_from_global(c[..., None], (0, 0, 0)); uc_mf.append(m)
        sf = list(nf); sf[d] += 1
        f = rng.standard_normal(tuple(sf))
        UF.append(f)
        # global array with one ghost layer around the whole fine index space (ghost values are overwritten)
        G = np.full(tuple(v + 2 for v in sf) + (1,), 1.0e30)
        G[1:-1, 1:-1, 1:-1, 0] = f
        m = lib.MultiFab(fine_lay, lib.face(d), 1, 1); m.set_from_global(G, (-1, -1, -1))
        # the valid faces keep f, ghosts get a sentinel
        uf_mf.append(m)
    for d in range(3):          # sentinel in the ghost faces (set_from_global copied neighbouring fine-space values there)
        for li in range(uf_mf[d].nlocal()):
            a, lo = uf_mf[d].to_numpy(li)
            blo, bhi, _ = fine_lay.local_box(li)
            b = np.full(a.shape, 7.0e30)
            sl = tuple(slice(blo[e] - lo[e], bhi[e] + (1 if e == d else 0) - lo[e] + 1) for e in range(3))
            b[sl] = a[sl]
            uf_mf[d].from_numpy(b, li)
    lib.create_umac_grown(uf_mf, uc_mf, cg, fg, ratio)
    # ---- oracle on the union box
    ucf, uff = [], []
    for d in range(3):
        cf = orc.Fab(nc, orc.face(d), 2, 1)
        idx = [np.mod(np.arange(cf.lo[e], cf.hi[e] + 1), nc[e]) for e in range(3)]
        # periodic images of a face array: face index f and f + n are the same face
        fi = np.arange(cf.lo[d], cf.hi[d] + 1)
        idx[d] = np.where(fi < 0, fi + nc[d], np.where(fi > nc[d], fi - nc[d], fi))
        cf.a[..., 0] = UC[d][np.ix_(*idx)]
        ucf.append(cf)
        lo = [vlo[e] - 1 for e in range(3)]; hi = [vhi[e] + 1 + (1 if e == d else 0) for e in range(3)]
        ff = orc.Fab(nf, orc.face(d), 0, 1, lo=lo, hi=hi)
        ff.a[...] = 7.0e30
        ff.a[1:-1, 1:-1, 1:-1, 0] = UF[d][vlo[0]:vhi[0] + 1 + (d == 0), vlo[1]:vhi[1] + 1 + (d == 1), vlo[2]:vhi[2] + 1 + (d == 2)]
        uff.append(ff)
    fdx = (C.c_double * 3)(*[1.0 / nf[e] for e in range(3)])
    L.orc_create_umac_grown(orc.fabptrs(uff), orc.i3(vlo), orc.i3(vhi), orc.fabptrs(ucf), ratio, fdx, orc.i3(nf), orc.i3((1, 1, 1)))
    for d in range(3):
        for li in range(uf_mf[d].nlocal()):
            a, lo = uf_mf[d].to_numpy(li)
            sl = tuple(slice(lo[e] - uff[d].lo[e], lo[e] - uff[d].lo[e] + a.shape[e]) for e in range(3))
            ref = uff[d].a[sl]
            # grid edges / corners of the union stay unset in the oracle only where the GPU array also holds interpolated data; compare
            # everywhere (both sides interpolate all ghost faces)
            assert np.abs(ref).max() < 1e3 and np.abs(a).max() < 1e3
            assert np.array_equal(a, ref), (d, li, np.abs(a - ref).max())
    # divergence of the corrected field in the face-adjacent ghost cells of the union
    u, v, w = (uff[d].a[..., 0] for d in range(3))
    dxf = 1.0 / nf[0]
    div = (u[1:, :, :] - u[:-1, :, :] + v[:, 1:, :] - v[:, :-1, :] + w[:, :, 1:] - w[:, :, :-1]) / dxf
    assert np.abs(div[0, 1:-1, 1:-1]).max() < 1e-10 and np.abs(div[-1, 1:-1, 1:-1]).max() < 1e-10
    assert np.abs(div[1:-1, 0, 1:-1]).max() < 1e-10 and np.abs(div[1:-1, 1:-1, -1]).max() < 1e-10
    assert np.abs(div[1:-1, 1:-1, 1:-1]).max() > 1.0            # the (random) interior is not divergence free: the fix is what zeroed the ring


def test_mac_sync_solve_makes_the_composite_mac_field_divergence_free(gpu):
    """MacProj::mac_sync_solve (Source/MacProj.cpp:359-470) on a 2-level hierarchy: coarse MAC projection (periodic 16^3), fine MAC
    projection on a refined box with the coarse phi on its coarse/fine faces, mac register = fine - coarse face velocity on the
    interface (CrseInit -area, FineAdd +area).  The pin: the divergence of the coarse cells next to the fine grid, evaluated with the
    fine fluxes on the shared faces, is the register's reflux; adding the sync correction Ucorr as IAMR applies it (the coarse field
    advects with u_mac - ... see below) removes it to solver tolerance."""
    lib = gpu
    ncr, nf = 16, 32
    nc, n = (ncr,) * 3, (nf,) * 3
    gc, gf = lib.Geom.make(nc), lib.Geom.make(n)
    clay = lib.Layout.decompose(nc, 8)
    fboxes = [((8, 8, 8), (23, 23, 23))]
    flay = lib.Layout(fboxes)
    dt = 0.01

    def face_field(nn, d, seed):
        t = [1 if e == d else 0 for e in range(3)]
        ax = [(np.arange(-1, nn + t[e] + 1) + (0.0 if t[e] else 0.5)) / nn for e in range(3)]
        X, Y, Z = np.meshgrid(*ax, indexing="ij")
        ph = np.random.default_rng(seed).uniform(0, 2 * np.pi, 3)
        return (np.sin(2 * np.pi * X + ph[0]) * np.cos(2 * np.pi * Y + ph[1]) + 0.5 * np.cos(2 * np.pi * Z + ph[2]))[..., None]

    def rho_cc(nn):
        x = (np.arange(-1, nn + 1) + 0.5) / nn
        X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
        return (1.0 + 0.3 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y))[..., None]

    umc = [lib.MultiFab(clay, lib.face(d), 1, 1) for d in range(3)]
    umf = [lib.MultiFab(flay, lib.face(d), 1, 1) for d in range(3)]
    for d in range(3):
        umc[d].set_from_global(face_field(ncr, d, 10 + d), (-1, -1, -1))
        umf[d].set_from_global(face_field(nf, d, 10 + d) + 0.05 * face_field(nf, d, 40 + d), (-1, -1, -1))
    rhoc = lib.MultiFab(clay, lib.CELL, 1, 1); rhoc.set_from_global(rho_cc(ncr), (-1, -1, -1))
    rhof = lib.MultiFab(flay, lib.CELL, 1, 1); rhof.set_from_global(rho_cc(nf), (-1, -1, -1))
    phic = lib.MultiFab(clay, lib.CELL, 1, 1); phic.setval(0.0)
    phif = lib.MultiFab(flay, lib.CELL, 1, 1); phif.setval(0.0)
    lib.mlmg_mac_solve(gc, umc, rhoc, 0, None, phic, 2.0 / dt, mac_tol=1e-12)
    lib.mlmg_mac_solve_cf(gf, umf, rhof, 0, None, phif, 2.0 / dt, phic, gc, 2, mac_tol=1e-12)
    dc = 1.0 / ncr
    area_c, area_f, vol_c = dc * dc, (dc / 2) ** 2, dc ** 3
    mr = lib.FluxRegister(flay, clay, gc, 2, 1)
    mr.setVal(0.0)
    for d in range(3):
        mr.CrseInit(umc[d], d, 0, 0, 1, -area_c)
        mr.FineAdd(umf[d], d, 0, 0, 1, area_f)
    # composite divergence of the unsynced coarse field = reflux of the register (coarse field itself is divergence free)
    comp0 = lib.MultiFab(clay, lib.CELL, 1, 0)
    lib.mac_divergence(gc, comp0, umc)
    assert comp0.norm0() <= 1e-9
    mr.Reflux(comp0, vol_c, -1.0, 0, 0, 1)
    mismatch = comp0.norm0()
    assert mismatch > 1e-2                                   # the two levels do disagree on the interface
    ucorr = [lib.MultiFab(clay, lib.face(d), 1, 0) for d in range(3)]
    sync_phi = lib.MultiFab(clay, lib.CELL, 1, 1)
    st = lib.mac_sync_solve(gc, mr, rhoc, dt, flay, ucorr, sync_phi, tol=1e-11)
    assert st.converged
    # D(Ucorr) equals the interface mismatch on the cells outside the fine grid (and vanishes elsewhere): u_mac - Ucorr is the
    # coarse velocity consistent with the fine one, the field mac_sync_compute re-advects with
    dU = lib.MultiFab(clay, lib.CELL, 1, 0)
    lib.mac_divergence(gc, dU, ucorr)
    D = dU.gather_valid(nc)[..., 0]
    R = comp0.gather_valid(nc)[..., 0]
    under = np.zeros(nc, bool)
    under[4:12, 4:12, 4:12] = True
    assert np.abs(D - R)[~under].max() <= 1e-8 * mismatch
    assert np.abs(D[under]).max() <= 1e-8 * mismatch
