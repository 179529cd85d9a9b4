"""CPU suite: the C-ABI shared library loads, exports every symbol declared in include/iamrx.h, fails loudly
without a device (no CPU fallback), and the host-only ghost-exchange planner is correct."""
import ctypes as C
import os
import re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "iamrx.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(iamrx_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from iamr_amd import lib
    L = lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 50
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_no_cpu_fallback_without_device():
    """on a box without a GPU iamrx_init must fail loudly; on a GPU box this test is vacuous"""
    from iamr_amd import lib
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(lib.IamrxError):
        lib.init(0)


def test_product_does_not_reference_the_oracle():
    """nothing under iamr_amd/ (sources or the built library) may import / link / dlopen the oracle"""
    for dp, _, files in os.walk(os.path.join(ROOT, "iamr_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "liborc" not in txt and "orc_" not in txt and "oracle/" not in txt, (dp, f)
    blob = open(os.path.join(ROOT, "iamr_amd", "libiamrx.so"), "rb").read()
    assert b"liborc" not in blob and b"orc_abec" not in blob


def host_plan(lib, boxes, owners, rank, typ, ng, geom):
    L = lib.lib()
    nb = len(boxes)
    arr = (C.c_int * (6 * nb))()
    for i, (lo, hi) in enumerate(boxes):
        for d in range(3):
            arr[6 * i + d] = lo[d]
            arr[6 * i + 3 + d] = hi[d]
    own = (C.c_int * nb)(*owners)
    n = C.c_int()
    lib.check(L.iamrx_host_fill_plan(nb, arr, own, rank, lib.i3(typ), ng, C.byref(geom), 0, None, C.byref(n)))
    desc = (C.c_int * (16 * n.value))()
    lib.check(L.iamrx_host_fill_plan(nb, arr, own, rank, lib.i3(typ), ng, C.byref(geom), n.value, desc, C.byref(n)))
    return np.array(desc[:], dtype=np.int64).reshape(n.value, 16)


def apply_plan_numpy(desc, fabs, los):
    """execute local descriptors on numpy fabs (dict global box -> array, origin)"""
    for d in desc:
        if d[0] != 0:
            continue
        s, t = int(d[2]), int(d[3])
        lo, hi, sh = d[4:7], d[7:10], d[10:13]
        dst = tuple(slice(lo[q] - los[t][q], hi[q] - los[t][q] + 1) for q in range(3))
        src = tuple(slice(lo[q] + sh[q] - los[s][q], hi[q] + sh[q] - los[s][q] + 1) for q in range(3))
        fabs[t][dst] = fabs[s][src]


@pytest.mark.parametrize("typ,ng", [((0, 0, 0), 1), ((0, 0, 0), 3), ((1, 1, 1), 1), ((1, 0, 0), 1)])
def test_host_fill_plan_single_rank_periodic(typ, ng):
    """ghost cells filled by the plan equal the periodic image of a global field (multi-box level)"""
    from iamr_amd import lib
    n = (16, 8, 8)
    geom = lib.Geom.make(n)
    boxes = [((i0, j0, 0), (i0 + 7, j0 + 3, 7)) for i0 in (0, 8) for j0 in (0, 4)]
    owners = [0] * len(boxes)
    desc = host_plan(lib, boxes, owners, 0, typ, ng, geom)
    rng = np.random.default_rng(0)
    G = rng.standard_normal(n)                      # periodic global field on the owner copies

    def gval(I, J, K):
        return G[np.mod(I, n[0]), np.mod(J, n[1]), np.mod(K, n[2])]
    fabs, los = {}, {}
    for b, (lo, hi) in enumerate(boxes):
        flo = [lo[d] - ng for d in range(3)]
        fhi = [hi[d] + typ[d] + ng for d in range(3)]
        I, J, K = np.meshgrid(*[np.arange(flo[d], fhi[d] + 1) for d in range(3)], indexing="ij")
        a = np.full(I.shape, np.nan)
        v = tuple(slice(ng, ng + hi[d] - lo[d] + 1 + typ[d]) for d in range(3))
        a[v] = gval(I[v], J[v], K[v])
        fabs[b], los[b] = a, flo
    apply_plan_numpy(desc, fabs, los)
    for b, (lo, hi) in enumerate(boxes):
        flo = los[b]
        I, J, K = np.meshgrid(*[np.arange(flo[d], flo[d] + fabs[b].shape[d]) for d in range(3)], indexing="ij")
        assert not np.isnan(fabs[b]).any()
        assert np.array_equal(fabs[b], gval(I, J, K))


def test_host_fill_plan_two_ranks_messages_match():
    """the pack list of rank 0 -> 1 and the unpack list of rank 1 <- 0 describe the same points in the same order"""
    from iamr_amd import lib
    n = (16, 8, 8)
    geom = lib.Geom.make(n)
    boxes = [((i0, 0, 0), (i0 + 3, 7, 7)) for i0 in (0, 4, 8, 12)]
    owners = [0, 0, 1, 1]
    p0 = host_plan(lib, boxes, owners, 0, (0, 0, 0), 2, geom)
    p1 = host_plan(lib, boxes, owners, 1, (0, 0, 0), 2, geom)
    send01 = p0[(p0[:, 0] == 1) & (p0[:, 1] == 1)]
    recv10 = p1[(p1[:, 0] == 2) & (p1[:, 1] == 0)]
    assert len(send01) == len(recv10) > 0
    assert np.array_equal(send01[:, 4:14], recv10[:, 4:14])       # same regions, shifts and buffer offsets
    send10 = p1[(p1[:, 0] == 1) & (p1[:, 1] == 0)]
    recv01 = p0[(p0[:, 0] == 2) & (p0[:, 1] == 1)]
    assert np.array_equal(send10[:, 4:14], recv01[:, 4:14])


@pytest.mark.parametrize("n,boxes,ng", [((16, 2, 8), [((0, 0, 0), (15, 1, 7))], 4),                                     # a two-cell slab level, four ghost layers
                                        ((16, 8, 8), [((i0, j0, 0), (i0 + 7, j0 + 3, 7)) for i0 in (0, 8) for j0 in (0, 4)], 2),
                                        ((8, 4, 4), [((0, 0, 0), (7, 3, 3))], 4)])
def test_host_fill_plan_gives_every_nodal_ghost_point_one_local_source(n, boxes, ng):
    """round 5: nodal boxes share the points on their faces and a periodic direction narrower than the ghost width maps a box onto a ghost
    point under one AND under two periods -- two local descriptors writing one ghost point in one launch, with sources that may differ in
    the last bit (the duplicates of a periodic node): a write-write race (found on the slab levels, tests/test_gpu_slab_mg.py).  The plan
    now keeps one local source per ghost point: no two local descriptors of a destination box overlap, every ghost point is still
    covered, and the values are the periodic images."""
    from iamr_amd import lib
    typ = (1, 1, 1)
    geom = lib.Geom.make(n)
    desc = host_plan(lib, boxes, [0] * len(boxes), 0, typ, ng, geom)
    loc = desc[desc[:, 0] == 0]
    count = {}
    for b, (lo, hi) in enumerate(boxes):
        flo = [lo[d] - ng for d in range(3)]
        shape = [hi[d] - lo[d] + 1 + typ[d] + 2 * ng for d in range(3)]
        count[b] = (np.zeros(shape, dtype=np.int64), flo)
    for d in loc:
        t = int(d[3])
        c, flo = count[t]
        sl = tuple(slice(int(d[4 + q]) - flo[q], int(d[7 + q]) - flo[q] + 1) for q in range(3))
        c[sl] += 1
    for b, (lo, hi) in enumerate(boxes):
        c, flo = count[b]
        valid = tuple(slice(ng, ng + hi[d] - lo[d] + 1 + typ[d]) for d in range(3))
        assert c[valid].max() == 0                     # nothing is copied onto a box's own valid points
        ghost = np.ones(c.shape, dtype=bool); ghost[valid] = False
        assert c[ghost].min() == 1 and c[ghost].max() == 1, (int(c[ghost].min()), int(c[ghost].max()))
    # and the values: periodic images of a node field whose duplicates agree
    rng = np.random.default_rng(1)
    G = rng.standard_normal(n)

    def gval(I, J, K):
        return G[np.mod(I, n[0]), np.mod(J, n[1]), np.mod(K, n[2])]
    fabs, los = {}, {}
    for b, (lo, hi) in enumerate(boxes):
        flo = [lo[d] - ng for d in range(3)]
        I, J, K = np.meshgrid(*[np.arange(flo[d], hi[d] + typ[d] + ng + 1) for d in range(3)], indexing="ij")
        a = np.full(I.shape, np.nan)
        v = tuple(slice(ng, ng + hi[d] - lo[d] + 1 + typ[d]) for d in range(3))
        a[v] = gval(I[v], J[v], K[v])
        fabs[b], los[b] = a, flo
    apply_plan_numpy(desc, fabs, los)
    for b in fabs:
        flo = los[b]
        I, J, K = np.meshgrid(*[np.arange(flo[d], flo[d] + fabs[b].shape[d]) for d in range(3)], indexing="ij")
        assert np.array_equal(fabs[b], gval(I, J, K))


def host_plan_ext(lib, boxes, owners, rank, typ, ng, geom, wall_ext):
    L = lib.lib()
    nb = len(boxes)
    arr = (C.c_int * (6 * nb))()
    for i, (lo, hi) in enumerate(boxes):
        for d in range(3):
            arr[6 * i + d] = lo[d]
            arr[6 * i + 3 + d] = hi[d]
    own = (C.c_int * nb)(*owners)
    n = C.c_int()
    lib.check(L.iamrx_host_fill_plan_wall_ext(nb, arr, own, rank, lib.i3(typ), ng, C.byref(geom), wall_ext, 0, None, C.byref(n)))
    desc = (C.c_int * (16 * n.value))()
    lib.check(L.iamrx_host_fill_plan_wall_ext(nb, arr, own, rank, lib.i3(typ), ng, C.byref(geom), wall_ext, n.value, desc, C.byref(n)))
    return np.array(desc[:], dtype=np.int64).reshape(n.value, 16)


@pytest.mark.parametrize("owners", [[0, 1, 1, 0], [0, 1, 0, 1], [0, 1, 2, 1], [0, 0, 1, 1]])
@pytest.mark.parametrize("n,ng", [((16, 8, 8), 2), ((8, 8, 2), 4)])
def test_host_fill_plan_gives_every_nodal_ghost_point_one_source_across_ranks(owners, n, ng):
    """round 6 (VERDICT r5 weak 12 / DESIGN round-5 'next' list): the one-source rule also covers REMOTE sources.  Two boxes of one peer
    (or of two peers) share the nodes on their common face, and a periodic direction narrower than the ghost width maps one remote box
    onto a ghost node twice: both copies used to arrive in the peer's unpack launch(es).  Over all ranks every ghost node of a box has
    exactly one descriptor (local copy or unpack), own-rank sources go first, and each pack list equals the peer's unpack list."""
    from iamr_amd import lib
    typ = (1, 1, 1)
    geom = lib.Geom.make(n)
    hx, hy = n[0] // 2, n[1] // 2
    boxes = [((i0, j0, 0), (i0 + hx - 1, j0 + hy - 1, n[2] - 1)) for i0 in (0, hx) for j0 in (0, hy)]
    nr = max(owners) + 1
    plans = [host_plan_ext(lib, boxes, owners, r, typ, ng, geom, 0) for r in range(nr)]
    for b, (lo, hi) in enumerate(boxes):
        r = owners[b]
        flo = [lo[d] - ng for d in range(3)]
        shape = [hi[d] - lo[d] + 1 + typ[d] + 2 * ng for d in range(3)]
        c = np.zeros(shape, dtype=np.int64)
        p = plans[r]
        for d in p[((p[:, 0] == 0) | (p[:, 0] == 2)) & (p[:, 3] == b)]:
            sl = tuple(slice(int(d[4 + q]) - flo[q], int(d[7 + q]) - flo[q] + 1) for q in range(3))
            c[sl] += 1
        valid = tuple(slice(ng, ng + hi[d] - lo[d] + 1 + typ[d]) for d in range(3))
        assert c[valid].max() == 0
        ghost = np.ones(c.shape, dtype=bool); ghost[valid] = False
        assert c[ghost].min() == 1 and c[ghost].max() == 1, (b, int(c[ghost].min()), int(c[ghost].max()))
    for a in range(nr):
        for b in range(nr):
            if a == b:
                continue
            send = plans[a][(plans[a][:, 0] == 1) & (plans[a][:, 1] == b)]
            recv = plans[b][(plans[b][:, 0] == 2) & (plans[b][:, 1] == a)]
            assert len(send) == len(recv)
            assert np.array_equal(send[:, 4:14], recv[:, 4:14])
    # a node a box of the destination's own rank can supply never travels
    for r in range(nr):
        p = plans[r]
        for d in p[p[:, 0] == 2]:
            b = int(d[3]); lo = d[4:7]; hi = d[7:10]
            for s, (slo, shi) in enumerate(boxes):
                if owners[s] != r:
                    continue
                for sx in (-1, 0, 1):
                    for sy in (-1, 0, 1):
                        for sz in range(-ng, ng + 1):
                            if s == b and sx == sy == sz == 0:
                                continue
                            sh = (sx * n[0], sy * n[1], sz * n[2])
                            inter = all(max(lo[q], slo[q] + sh[q]) <= min(hi[q], shi[q] + typ[q] + sh[q]) for q in range(3))
                            assert not inter, (r, b, s, sh)


@pytest.mark.parametrize("owners", [[0, 0, 0, 0], [0, 1, 1, 0]])
def test_host_fill_plan_wall_ext_hands_on_the_ghost_cells_beyond_walls(owners):
    """round 6: the density copy of the multi-box red + black sweep (CellMG::m_sig2) needs, in its EDGE ghost cells
    beyond a domain wall behind a box-box face, what the box next door holds in its face ghost cells there.  With wall_ext = 1 the source
    boxes of the plan reach one cell beyond the non-periodic sides they touch: executed on fabs whose face ghost cells beyond the walls
    carry a boundary fill F and whose other ghost cells are NaN, every ghost cell that has a box (or a periodic image of one) beside it
    in-plane ends up with F of the global field -- only the cells beyond TWO walls and the own face ghosts are left alone."""
    from iamr_amd import lib
    n = (16, 8, 8)
    geom = lib.Geom.make(n, periodic=(1, 0, 0))
    boxes = [((i0, j0, 0), (i0 + 7, j0 + 3, 7)) for i0 in (0, 8) for j0 in (0, 4)]
    ng = 2
    typ = (0, 0, 0)
    rng = np.random.default_rng(2)
    E = rng.standard_normal((n[0], n[1] + 2, n[2] + 2))          # the field incl. one layer beyond the y / z walls

    def gval(I, J, K):
        return E[np.mod(I, n[0]), J + 1, K + 1]
    nr = max(owners) + 1
    fabs, los = {}, {}
    for b, (lo, hi) in enumerate(boxes):
        flo = [lo[d] - ng for d in range(3)]
        I, J, K = np.meshgrid(*[np.arange(flo[d], hi[d] + ng + 1) for d in range(3)], indexing="ij")
        a = np.full(I.shape, np.nan)
        v = tuple(slice(ng, ng + hi[d] - lo[d] + 1) for d in range(3))
        a[v] = gval(I[v], J[v], K[v])
        # the owner's boundary fill: the first layer beyond the walls this box touches, over its own footprint only
        for d in (1, 2):
            for side in (0, 1):
                if (lo[d] if side == 0 else hi[d]) != (0 if side == 0 else n[d] - 1):
                    continue
                sl = list(v)
                sl[d] = slice(ng - 1, ng) if side == 0 else slice(ng + hi[d] - lo[d] + 1, ng + hi[d] - lo[d] + 2)
                sl = tuple(sl)
                a[sl] = gval(I[sl], J[sl], K[sl])
        fabs[b], los[b] = a, flo
    for r in range(nr):
        desc = host_plan_ext(lib, boxes, owners, r, typ, ng, geom, 1)
        if nr == 1:
            apply_plan_numpy(desc, fabs, los)
        else:
            # emulate the exchange: pack of every rank first, then unpack
            bufs = {}
            for rr in range(nr):
                p = host_plan_ext(lib, boxes, owners, rr, typ, ng, geom, 1)
                for d in p[p[:, 0] == 1]:
                    s = int(d[2]); lo_, hi_, sh = d[4:7], d[7:10], d[10:13]
                    src = tuple(slice(lo_[q] + sh[q] - los[s][q], hi_[q] + sh[q] - los[s][q] + 1) for q in range(3))
                    bufs[(rr, int(d[1]), int(d[13]))] = fabs[s][src].copy()
            for rr in range(nr):
                p = host_plan_ext(lib, boxes, owners, rr, typ, ng, geom, 1)
                apply_plan_numpy(p, fabs, los)
                for d in p[p[:, 0] == 2]:
                    t = int(d[3]); lo_, hi_ = d[4:7], d[7:10]
                    dst = tuple(slice(lo_[q] - los[t][q], hi_[q] - los[t][q] + 1) for q in range(3))
                    fabs[t][dst] = bufs[(int(d[1]), rr, int(d[13]))]
            break
    for b, (lo, hi) in enumerate(boxes):
        flo = los[b]
        I, J, K = np.meshgrid(*[np.arange(flo[d], flo[d] + fabs[b].shape[d]) for d in range(3)], indexing="ij")
        out_y = (J < 0) | (J >= n[1]); out_z = (K < 0) | (K >= n[2])
        far = (J < -1) | (J > n[1]) | (K < -1) | (K > n[2])               # second layer beyond a wall: nobody's
        expect = ~(out_y & out_z) & ~far
        assert not np.isnan(fabs[b][expect]).any(), b
        assert np.array_equal(fabs[b][expect], gval(I[expect], np.clip(J, -1, n[1])[expect], np.clip(K, -1, n[2])[expect]))
        assert np.isnan(fabs[b][(out_y & out_z) | far]).all()
