"""GPU: the fused red+black GSRB sweep (one z-marching LDS pass + black pass on the box surfaces, out of place) is bit-identical
to the two colour passes with a ghost/BC fill in front of each, on single- and multi-box levels, periodic / Neumann / Dirichlet
(order 2) domains, scalar and 3-component (tensor-like) coefficient sets."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,boxes,per,bct,ncomp", [
    ((32, 32, 32), None, (1, 1, 1), 0, 1),
    ((96, 40, 36), None, (1, 1, 1), 0, 1),
    ((64, 32, 32), (32, 16, 16), (1, 1, 1), 0, 1),
    ((64, 48, 32), None, (0, 1, 0), 102, 1),
    ((64, 48, 32), (32, 24, 16), (0, 0, 0), 101, 1),
    ((48, 32, 32), (24, 16, 16), (0, 1, 0), 101, 3),
    ((192, 160, 96), None, (1, 1, 1), 0, 1),
])
def test_fused_sweep_equals_two_colour_passes(gpu, n, boxes, per, bct, ncomp):
    lib = gpu
    g = lib.Geom.make(n, periodic=per)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    rng = np.random.default_rng(3)
    lobc = tuple(0 if per[d] else bct for d in range(3))
    hibc = tuple(0 if per[d] else (101 if bct == 101 else 102) for d in range(3))
    b = []
    for d in range(3):
        nf = list(n); nf[d] += 1
        G = 0.5 + rng.random(tuple(nf) + (ncomp,))
        # periodic consistency of the face coefficients
        if per[d]:
            sl_lo = [slice(None)] * 3; sl_hi = [slice(None)] * 3
            sl_lo[d] = 0; sl_hi[d] = n[d]
            G[tuple(sl_hi)] = G[tuple(sl_lo)]
        m = lib.MultiFab(lay, lib.face(d), ncomp, 0)
        m.set_from_global(G, (0, 0, 0))
        b.append(m)
    A = 0.5 + rng.random(tuple(n) + (1,))
    a = lib.MultiFab(lay, lib.CELL, 1, 0); a.set_from_global(A, (0, 0, 0))
    Pg = np.zeros(tuple(v + 2 for v in n) + (ncomp,))
    Pg[1:-1, 1:-1, 1:-1] = rng.standard_normal(tuple(n) + (ncomp,))
    Rg = rng.standard_normal(tuple(n) + (ncomp,))
    res = []
    for fused in (0, 1):
        phi = lib.MultiFab(lay, lib.CELL, ncomp, 1); phi.set_from_global(Pg, (-1, -1, -1))
        rhs = lib.MultiFab(lay, lib.CELL, ncomp, 0); rhs.set_from_global(Rg, (0, 0, 0))
        for _ in range(2):
            lib.abec_gsrb_sweep(g, 1.3, 0.7, a, b, phi, rhs, 1.15, lobc, hibc, 2, fused)
        res.append(phi.gather_valid(n))
    assert np.abs(res[0] - Pg[1:-1, 1:-1, 1:-1]).max() > 1e-3
    assert np.array_equal(res[0], res[1]), np.abs(res[0] - res[1]).max()
