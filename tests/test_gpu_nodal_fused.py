"""GPU: the plane-fused nodal Gauss-Seidel (2 passes, LDS-staged, recomputed halo) is bit-identical to the
8 sequential colour passes, on single- and multi-box periodic levels, including boxes narrower than the tile."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


# the 160x128x96 case has far more workgroups than fit on the device at once: an in-place pass would let late workgroups
# read already-updated halo nodes (caught here as a mismatch with the sequential colour passes)
@pytest.mark.parametrize("n,boxes", [((16, 16, 16), None), ((48, 24, 20), None), ((32, 32, 16), (16, 16, 8)), ((4, 4, 4), None), ((2, 2, 2), None),
                                     ((160, 128, 96), None), ((128, 128, 64), (64, 64, 32))])
def test_fused_sweep_equals_eight_colour_passes(gpu, n, boxes):
    lib = gpu
    from iamr_amd import ns as N
    g = lib.Geom.make(n)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    rng = np.random.default_rng(5)
    # periodic global fields on the owner copies
    Xg = rng.standard_normal(n)
    Rg = rng.standard_normal(n)
    Sg = 1.0 / (1.0 + 0.8 * rng.random(n))

    def node_field(G, ng):
        idx = [np.mod(np.arange(-ng, n[d] + 1 + ng), n[d]) for d in range(3)]
        return G[np.ix_(*idx)][..., None]

    def cell_field(G, ng):
        idx = [np.mod(np.arange(-ng, n[d] + ng), n[d]) for d in range(3)]
        return G[np.ix_(*idx)][..., None]

    res = []
    modes = [(0, 4), (1, 4)]
    if boxes is None and n[0] * n[1] * n[2] <= 8 ** 3:
        modes.append((2, 1))       # single-workgroup coarse-level smoother
    for fused, ng in modes:
        x = lib.MultiFab(lay, lib.NODE, 1, ng)
        r = lib.MultiFab(lay, lib.NODE, 1, ng)
        s = lib.MultiFab(lay, lib.CELL, 1, ng)
        x.set_from_global(node_field(Xg, ng), (-ng,) * 3)
        r.set_from_global(node_field(Rg, ng), (-ng,) * 3)
        s.set_from_global(cell_field(Sg, ng), (-ng,) * 3)
        for _ in range(2):
            N.nodal_gs_sweep(g, x, r, s, fused)
        x.fill_boundary(g)
        res.append(x.gather_valid(n))
    for q in range(1, len(res)):
        assert np.array_equal(res[0], res[q]), (modes[q], np.abs(res[0] - res[q]).max())
