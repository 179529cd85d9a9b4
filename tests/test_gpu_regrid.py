"""error estimation and grid generation of one AMR level on the device -- the `mag_vort` derive (dermgvort,
Source/NS_derive.cpp:86-264), the AMRErrorTag tests of NavierStokes::error_setup / errorEst (Source/NS_error.cpp:10-145) against
the oracle, the Berger-Rigoutsos grid generation (properties: every tagged cell covered, boxes disjoint / aligned / bounded,
efficiency), and NavierStokesBase::init(old) for the new level (FillPatch: old fine data where it exists, conservative-linear
interpolation of the coarse level elsewhere)."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def vortex_field(n, ng=1):
    x = [(np.arange(-ng, n[d] + ng) + 0.5) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*x, indexing="ij")
    r2 = (X - 0.4) ** 2 + (Y - 0.55) ** 2
    u = -(Y - 0.55) * np.exp(-60 * r2) * (1 + 0.3 * np.sin(2 * np.pi * Z))
    v = (X - 0.4) * np.exp(-60 * r2) * (1 + 0.3 * np.sin(2 * np.pi * Z))
    w = 0.05 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y)
    tr = np.exp(-80 * ((X - 0.6) ** 2 + (Y - 0.3) ** 2 + (Z - 0.5) ** 2))
    return u, v, w, tr


@pytest.mark.parametrize("boxes", [None, 16])
def test_mag_vort_and_error_tags_match_oracle(orc, gpu, boxes):
    lib = gpu
    L = orc.lib()
    n = (32, 32, 16)
    g_o, g_d = orc.geom(n), lib.Geom.make(n)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    u, v, w, tr = vortex_field(n)
    vel = orc.Fab(n, orc.CELL, 1, 4)
    for c, a in enumerate((u, v, w, tr)):
        vel.a[..., c] = a
    L.orc_fill_periodic(vel.ref(), C.byref(g_o), orc.i3(orc.CELL))
    vort_o = orc.Fab(n, orc.CELL, 0, 1)
    L.orc_mag_vort(C.byref(g_o), vort_o.ref(), vel.ref())
    vel_d = lib.MultiFab(lay, lib.CELL, 4, 1); vel_d.set_from_global(vel.a, vel.lo)
    vort_d = lib.MultiFab(lay, lib.CELL, 1, 0)
    lib.derive_mag_vort(g_d, vort_d, vel_d)
    assert np.array_equal(vort_d.gather_valid(n)[..., 0], vort_o.a[..., 0])
    # the four indicator types, accumulated into one tag array like errorEst does
    vmax = vort_o.a.max()
    specs = [(lib.TAG_VORT, vort_d, vort_o, 0, 0.25 * vmax, 1, None), (lib.TAG_GREATER, vel_d, vel, 3, 0.5, 0, None),
             (lib.TAG_LESS, vel_d, vel, 0, -0.04, 0, ((0.0, 0.0, 0.0), (1.0, 0.5, 1.0))), (lib.TAG_GRAD, vel_d, vel, 3, 0.08, 0, None)]
    tags_o = orc.Fab(n, orc.CELL, 0, 1)
    tags_d = lib.MultiFab(lay, lib.CELL, 1, 0); tags_d.setval(0.0)
    for mode, fd, fo, comp, val, lev, rb in specs:
        lo = (C.c_double * 3)(*rb[0]) if rb else None
        hi = (C.c_double * 3)(*rb[1]) if rb else None
        L.orc_error_tag(C.byref(g_o), tags_o.ref(), fo.ref(), comp, mode, C.c_double(val), lev, lo, hi)
        lib.error_tag(g_d, tags_d, fd, mode, val, comp=comp, level=lev, realbox=rb)
    T = tags_d.gather_valid(n)[..., 0]
    assert np.array_equal(T, tags_o.a[..., 0])
    assert 0 < T.sum() < 0.3 * T.size


def test_grid_generation_and_new_level_fill(orc, gpu):
    lib = gpu
    n = (32, 32, 16)
    g_d = lib.Geom.make(n)
    lay = lib.Layout.decompose(n, 16)
    u, v, w, tr = vortex_field(n)
    S = np.stack([u, v, w, tr], axis=-1)
    S_d = lib.MultiFab(lay, lib.CELL, 4, 1); S_d.set_from_global(S, (-1, -1, -1))
    S_d.fill_boundary(g_d)
    vort = lib.MultiFab(lay, lib.CELL, 1, 0)
    lib.derive_mag_vort(g_d, vort, S_d)
    tags = lib.MultiFab(lay, lib.CELL, 1, 0); tags.setval(0.0)
    lib.error_tag(g_d, tags, vort, lib.TAG_VORT, 0.3 * vort.norm0())
    lib.error_tag(g_d, tags, S_d, lib.TAG_GREATER, 0.5, comp=3)
    T = tags.gather_valid(n)[..., 0] > 0
    bf, mgs, eff, nbuf = 4, 16, 0.7, 1
    bx = lib.cluster_tags(g_d, tags, blocking_factor=bf, max_grid_size=mgs, grid_eff=eff, n_error_buf=nbuf)
    assert len(bx) >= 2
    cover = np.zeros(n, int)
    for lo, hi in bx:
        for d in range(3):
            assert lo[d] % bf == 0 and (hi[d] + 1) % bf == 0 and hi[d] - lo[d] + 1 <= mgs and 0 <= lo[d] <= hi[d] < n[d]
        cover[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1] += 1
    assert cover.max() == 1                                               # disjoint
    # every tagged cell, grown by n_error_buf (clipped at the domain), is covered
    Tb = T.copy()
    for d in range(3):
        for s in (-1, 1):
            sh = np.roll(T, s, axis=d)
            idx = [slice(None)] * 3
            idx[d] = 0 if s == 1 else -1
            sh[tuple(idx)] = False
            Tb |= sh
    assert np.all(cover[Tb] == 1)
    assert T.sum() / cover.sum() > 0.1 and cover.sum() < 0.5 * T.size         # refined region: a modest part of the domain (buffer + blocking factor 4)
    # the new level (refinement ratio 2) and its data -- NavierStokesBase::init(old): FillPatch from the old fine level (one box
    # here, holding recognisable data) where it overlaps, conservative-linear interpolation of the coarse state elsewhere
    fboxes = [(tuple(2 * q for q in lo), tuple(2 * q + 1 for q in hi)) for lo, hi in bx]
    nf = tuple(2 * q for q in n)
    flay = lib.Layout(fboxes)
    gf = lib.Geom.make(nf)
    old_box = ((16, 24, 0), (39, 47, 31))
    olay = lib.Layout([old_box])
    old = lib.MultiFab(olay, lib.CELL, 4, 0)
    old.setval(7.5)
    Sf = lib.MultiFab(flay, lib.CELL, 4, 0)
    lib.fillpatch_two_levels(Sf, 0.0, (None, old, 0.0, 0.0), (None, S_d, 0.0, 0.0), g_d, gf, ncomp=4)
    Sv = S[1:-1, 1:-1, 1:-1, :]
    n_old = n_new = 0
    for li in range(Sf.nlocal()):
        a, lo = Sf.to_numpy(li)
        blo, bhi, gi = flay.local_box(li)
        I, J, K = np.meshgrid(*[np.arange(blo[d], bhi[d] + 1) for d in range(3)], indexing="ij")
        in_old = (I >= old_box[0][0]) & (I <= old_box[1][0]) & (J >= old_box[0][1]) & (J <= old_box[1][1]) & (K >= old_box[0][2]) & (K <= old_box[1][2])
        assert np.all(a[in_old] == 7.5)                                          # old fine data are kept
        n_old += in_old.sum()
        # conservative interpolation: where a whole coarse cell is new, the mean of its 8 fine cells is the coarse value
        avg = a.reshape(a.shape[0] // 2, 2, a.shape[1] // 2, 2, a.shape[2] // 2, 2, 4).mean(axis=(1, 3, 5))
        newc = ~in_old.reshape(a.shape[0] // 2, 2, a.shape[1] // 2, 2, a.shape[2] // 2, 2).any(axis=(1, 3, 5))
        ref = Sv[blo[0] // 2:bhi[0] // 2 + 1, blo[1] // 2:bhi[1] // 2 + 1, blo[2] // 2:bhi[2] // 2 + 1, :]
        if newc.any():
            assert np.abs(avg - ref)[newc].max() <= 1e-13
        n_new += newc.sum()
    assert n_old > 0 and n_new > 0
