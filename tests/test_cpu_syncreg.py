"""The literal, box-by-box restatement of IAMR's SyncRegister (oracle/orc_syncreg.c, following Source/SyncRegister.cpp:18-607 line by
line) against (1) the closed form it is supposed to equal -- full weighting (1,2,1)^3/64 of the single-valued fine residual with even
reflection at walls, the statement oracle/orc_amr.c and the product's union-based register are built on -- on synthetic data, and
(2) the single-valued restatement inside complete multi-level oracle runs: abutting fine boxes (L shape), a fine box on a no-slip
wall, a fine box across a periodic boundary, three levels, and the reference's own Exec/run2d/test_grids/fixed_grids_{1..6} layouts
lifted to 3-D slabs.  No GPU: oracle against oracle and against numpy."""
import ctypes as C
import os
import numpy as np
import pytest

import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _L():
    L = orc.lib()
    for f in ("orc_ndmf_create", "orc_syncreg_create", "orc_ndmf_fab", "orc_syncreg_fab"):
        getattr(L, f).restype = C.c_void_p
    L.orc_syncreg_last_diff.restype = C.c_double
    return L


class NdMF:
    """nodal MultiFab of the oracle (orc_ndmf): numpy views of its fabs"""

    def __init__(self, boxes, ng):
        L = _L()
        flat = [v for lo, hi in boxes for v in (*lo, *hi)]
        self.h = C.c_void_p(L.orc_ndmf_create(len(boxes), (C.c_int * len(flat))(*flat), ng))
        self.boxes, self.ng, self.fabs = boxes, ng, []
        for b in range(len(boxes)):
            lo, hi = (C.c_int * 3)(), (C.c_int * 3)()
            p = L.orc_ndmf_fab(self.h, b, lo, hi)
            shape = tuple(hi[d] - lo[d] + 1 for d in range(3))
            a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_double)), shape=(shape[2], shape[1], shape[0])).T     # [i, j, k], i fastest
            self.fabs.append((a, list(lo)))

    def __del__(self):
        orc.lib().orc_ndmf_destroy(self.h)


def _surface_noise(rng, boxes, ng, dom_n=None, periodic=(0, 0, 0)):
    """per-box pieces of a fine sync residual: random on the surface nodes of every box's nodal box, zero inside and in the ghost nodes
    (what compSyncResidualFine leaves: a node inside a box sees all its cells)"""
    m = NdMF(boxes, ng)
    for (a, lo), (blo, bhi) in zip(m.fabs, boxes):
        a[...] = 0.0
        n = [bhi[d] - blo[d] + 2 for d in range(3)]
        v = rng.standard_normal(n)
        v[1:-1, 1:-1, 1:-1] = 0.0
        a[ng:ng + n[0], ng:ng + n[1], ng:ng + n[2]] = v
    return m


def _single_valued(m, n, periodic):
    """sum of the per-box pieces on the nodes [0, n] of the level (periodic duplicates identified)"""
    tot = np.zeros([n[d] + 1 for d in range(3)])
    for (a, lo), (blo, bhi) in zip(m.fabs, m.boxes):
        ng = m.ng
        sl = tuple(slice(blo[d], bhi[d] + 2) for d in range(3))
        tot[sl] += a[ng:a.shape[0] - ng, ng:a.shape[1] - ng, ng:a.shape[2] - ng]
    for d in range(3):
        if periodic[d]:
            s0 = [slice(None)] * 3; s1 = [slice(None)] * 3
            s0[d], s1[d] = 0, n[d]
            both = tot[tuple(s0)] + tot[tuple(s1)]
            tot[tuple(s0)] = both; tot[tuple(s1)] = both
    return tot


def _full_weighting(rf, nc, periodic):
    """(1,2,1)^3/64 restriction of the fine nodal field rf (nodes [0, 2 nc]) to the coarse nodes; ghost nodes: periodic images / even
    reflection about walls; at wall nodes the fine field is first doubled per wall (the doubled form of the wall rows)"""
    nf = [2 * c for c in nc]
    r = rf.copy()
    for d in range(3):
        if not periodic[d]:
            s0 = [slice(None)] * 3; s1 = [slice(None)] * 3
            s0[d], s1[d] = 0, nf[d]
            r[tuple(s0)] *= 2.0; r[tuple(s1)] *= 2.0
    g = np.zeros([nf[d] + 3 for d in range(3)])
    g[1:-1, 1:-1, 1:-1] = r
    for d in range(3):
        lo = [slice(None)] * 3; hi = [slice(None)] * 3; slo = [slice(None)] * 3; shi = [slice(None)] * 3
        lo[d], hi[d] = 0, nf[d] + 2
        if periodic[d]:
            slo[d], shi[d] = nf[d], 2            # node -1 = node nf-1 (array index nf), node nf+1 = node 1 (array index 2)
        else:
            slo[d], shi[d] = 2, nf[d]            # even reflection
        g[tuple(lo)] = g[tuple(slo)]; g[tuple(hi)] = g[tuple(shi)]
    w = np.array([0.25, 0.5, 0.25])
    out = np.zeros([nc[d] + 1 for d in range(3)])
    for a in range(3):
        for b in range(3):
            for c in range(3):
                out += w[a] * w[b] * w[c] * g[a:a + 2 * nc[0] + 1:2, b:b + 2 * nc[1] + 1:2, c:c + 2 * nc[2] + 1:2]
    return out


LAYOUTS = {
    "one box": ([8, 8, 8], (1, 1, 1), [([4, 4, 4], [11, 11, 11])]),
    "two abutting boxes": ([8, 8, 8], (1, 1, 1), [([2, 4, 4], [7, 11, 11]), ([8, 4, 4], [13, 11, 11])]),
    "L shape, three boxes": ([8, 8, 8], (1, 1, 1), [([2, 2, 4], [7, 7, 11]), ([8, 2, 4], [13, 7, 11]), ([2, 8, 4], [7, 13, 11])]),
    "box on a wall": ([8, 8, 8], (1, 1, 0), [([4, 4, 0], [11, 11, 7])]),
    "box in a wall corner": ([8, 8, 8], (0, 0, 1), [([0, 0, 4], [7, 7, 11])]),
    "box on the periodic boundary": ([8, 8, 8], (1, 1, 1), [([0, 4, 4], [7, 11, 11])]),
    "boxes on both sides of the periodic boundary": ([8, 8, 8], (1, 1, 1), [([0, 4, 4], [5, 11, 11]), ([10, 4, 4], [15, 11, 11])]),
    "periodic slab": ([8, 8, 4], (1, 1, 1), [([4, 4, 0], [11, 11, 7])]),
}


@pytest.mark.parametrize("name", list(LAYOUTS))
def test_fine_add_equals_full_weighting_of_the_single_valued_residual(name):
    """SyncRegister::FineAdd + InitRHS of the literal register on random per-box surface data = the closed form on every register node
    that is not surrounded by fine cells only (those carry noise the composite solve never reads; the 3-D maxcount of the reference
    does not mask them, SyncRegister.cpp:264-283)"""
    nc, per, boxes = LAYOUTS[name]
    L = _L()
    rng = np.random.default_rng(7)
    g = orc.geom(nc, periodic=per)
    fine = _surface_noise(rng, boxes, 1)
    flat = [v for lo, hi in boxes for v in (*lo, *hi)]
    sr = C.c_void_p(L.orc_syncreg_create(len(boxes), (C.c_int * len(flat))(*flat), 2))
    L.orc_syncreg_setval(sr, C.c_double(0.0))
    rf = _single_valued(fine, [2 * c for c in nc], per)
    before = [a.copy() for a, _ in fine.fabs]
    L.orc_syncreg_fine_add(sr, fine.h, C.byref(g), C.c_double(0.5))
    for (a, _), b in zip(fine.fabs, before):
        assert np.allclose(a, 0.5 * b, rtol=0, atol=1e-15)       # the edge / corner scaling is undone, only `mult` stays
    rhs = NdMF([([0, 0, 0], [c - 1 for c in nc])], 0)
    zero3 = (C.c_int * 3)(0, 0, 0)
    L.orc_syncreg_init_rhs(sr, rhs.h, C.byref(g), zero3, zero3)
    got = rhs.fabs[0][0]
    want = _full_weighting(0.5 * rf, nc, per)
    # node classes of the coarse level w.r.t. the union of the coarsened boxes
    cov = np.zeros([c + 2 for c in nc], dtype=int)
    for lo, hi in boxes:
        cov[tuple(slice(lo[d] // 2 + 1, hi[d] // 2 + 2) for d in range(3))] = 1
    for d in range(3):
        lo = [slice(None)] * 3; hi = [slice(None)] * 3; slo = [slice(None)] * 3; shi = [slice(None)] * 3
        lo[d], hi[d] = 0, nc[d] + 1
        slo[d], shi[d] = (nc[d], 1) if per[d] else (1, nc[d])
        cov[tuple(lo)] = cov[tuple(slo)]; cov[tuple(hi)] = cov[tuple(shi)]
    cnt = sum(cov[a:a + nc[0] + 1, b:b + nc[1] + 1, c:c + nc[2] + 1] for a in range(2) for b in range(2) for c in range(2))
    on_bndry = (cnt > 0) & (cnt < 8)
    assert on_bndry.any()
    assert abs(got - want)[on_bndry].max() < 1e-13 * max(1.0, abs(want).max()), name
    assert abs(got[cnt == 0]).max() == 0.0                        # nothing outside the registers
    L.orc_syncreg_destroy(sr)


def _grids_2d(path):
    """boxes of the first refined level of a reference grid file (coarse index space of the 2-D run), lifted to a slab"""
    lines = [t.split("#")[0].strip() for t in open(path).read().split("\n")]
    lines = [t for t in lines if t]
    nb = int(lines[1])
    out = []
    import re
    for q in range(nb):
        m = re.match(r"\(\(([-\d, ]+)\)\s*\(([-\d, ]+)\)", lines[2 + q])
        lo = [int(v) for v in m.group(1).split(",")]
        hi = [int(v) for v in m.group(2).split(",")]
        out.append((lo, hi))
    return out


def _run_case(n, per, fine, nsteps=1, **kw):
    L = _L()
    L.orc_syncreg_last_diff(1)
    a = orc.OrcAmr(orc.geom(n, periodic=per), orc.ns_params(**kw), orc.mg_opts(), [[]] + fine)
    for l in range(1 + len(fine)):
        X, Y, Z = a.cell_centres(l)
        S = orc.taylorgreen_state(X, Y, Z, c=1.0)
        for d in range(3):
            if not per[d]:
                S[..., d] = (0.3 + S[..., d]) * np.sin(np.pi * (X, Y, Z)[d])      # no flow through the walls (the initial projection does the rest)
        S[..., 4] = np.exp(-((X - 0.4) ** 2 + (Y - 0.55) ** 2 + (Z - 0.5) ** 2) / 0.03)
        a.set_state(l, S)
    a.post_init()
    for _ in range(nsteps):
        a.step()
        st = a.sync_stats()
        assert st.converged == 1
    return L.orc_syncreg_last_diff(1), a


CASES = {
    "L shape": ([8, 8, 8], (1, 1, 1), [[([4, 4, 4], [9, 9, 11]), ([10, 4, 4], [13, 9, 11]), ([4, 10, 4], [9, 13, 11])]], {}),
    "box on a no-slip wall": ([8, 8, 8], (1, 1, 0), [[([4, 4, 0], [11, 11, 7])]], dict(phys_lo=[0, 0, 5], phys_hi=[0, 0, 5], visc_coef=0.01)),
    "box across the periodic boundary": ([8, 8, 8], (1, 1, 1), [[([0, 4, 4], [5, 11, 11]), ([12, 4, 4], [15, 11, 11])]], {}),
    "three levels": ([8, 8, 8], (1, 1, 1), [[([2, 2, 2], [13, 13, 13])], [([10, 10, 10], [21, 21, 21])]], {}),
}


@pytest.mark.parametrize("name", list(CASES))
def test_literal_register_agrees_with_the_single_valued_one_inside_a_run(name):
    """complete multi-level oracle runs (post_init + a coarse step) drive both registers with the same level projections; the right-hand
    sides they hand to MLsyncProject agree to the level of the projections' own tolerance (the two differ only in how they treat the
    converged residual of interior nodes)"""
    n, per, fine, kw = CASES[name]
    diff, a = _run_case(n, per, fine, cfl=0.7, init_iter=1, **kw)
    assert diff < 1e-8, (name, diff)


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6])
def test_reference_fixed_grids_as_slabs(k):
    """Exec/run2d/test_grids/fixed_grids_k (the reference's own regression layouts for the SyncRegister path: abutting boxes, boxes on
    walls, boxes across the periodic boundary), first refined level, lifted to a 4-cell-deep periodic slab of the 3-D algorithm; the
    64 x 64 layouts are used at 1/4 of their resolution where every box stays aligned, otherwise as they are on 32 x 32"""
    path = os.path.join(GOLD, f"fixed_grids_{k}")
    boxes2 = _grids_2d(path)
    span = max(max(hi) for lo, hi in boxes2) + 1
    n0 = 16
    while n0 < span:
        n0 *= 2
    if n0 > 32:
        pytest.skip("fixed_grids_6 needs a 64 x 64 base grid: minutes on the CPU oracle (its level-0/1 layout is covered at 32 x 32 by sets 3 and 5)")
    fine = [([2 * lo[0], 2 * lo[1], 0], [2 * hi[0] + 1, 2 * hi[1] + 1, 7]) for lo, hi in boxes2]
    # periodicity of the reference's inputs_k (x periodic except set 2, y periodic for sets 4 and 5); its walls / inflow / outflow in y
    # become slip walls (the hotspot physics of sets 1-3 needs do_temp / divu)
    per = {1: (1, 0, 1), 2: (0, 0, 1), 3: (1, 0, 1), 4: (1, 1, 1), 5: (1, 1, 1), 6: (1, 0, 1)}[k]
    plo = [0 if per[d] else 4 for d in range(3)]
    diff, a = _run_case([n0, n0, 4], per, [fine], cfl=0.7, init_iter=1, phys_lo=plo, phys_hi=plo)
    assert diff < 1e-8, (k, diff)
