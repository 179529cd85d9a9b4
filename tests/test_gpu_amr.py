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
