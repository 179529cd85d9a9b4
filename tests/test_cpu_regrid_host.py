"""The host side of the grid generation (iamr_amd/csrc/regrid.hip cluster_tags, amrregrid.hip erode_map), through the host-array entries
of the C-ABI: no device is touched.  Round 6 rewrote the loops that walk the index space of a level (the nesting erosion without whole-map
copies, the tag buffer by block ranges, the allowed-block test eight cells at a time): these tests pin them on plain numpy restatements of
what they compute -- AmrMesh::MakeNewGrids' tag buffer + blocking-factor coarsening (TagBoxArray::buffer / coarsen), the proper-nesting
erosion of Amr::regrid(lbase > 0) -- and on the properties every box list must have."""
import ctypes as C
import numpy as np
import pytest


def _lib():
    from iamr_amd import lib
    return lib, lib.lib()


def _erode_ref(m, per, passes):
    """a cell survives a pass if all 26 neighbours are set; periodic images count, nothing constrains beyond a non-periodic face"""
    m = m.copy()
    for _ in range(passes):
        out = m.copy()
        for dk in (-1, 0, 1):
            for dj in (-1, 0, 1):
                for di in (-1, 0, 1):
                    if (di, dj, dk) == (0, 0, 0):
                        continue
                    nb = m
                    for ax, d in ((0, di), (1, dj), (2, dk)):      # m is indexed [i, j, k]
                        if d == 0:
                            continue
                        if per[ax]:
                            nb = np.roll(nb, -d, axis=ax)
                        else:
                            sh = np.ones_like(nb)
                            src = [slice(None)] * 3; dst = [slice(None)] * 3
                            if d > 0: src[ax] = slice(d, None); dst[ax] = slice(0, -d)
                            else: src[ax] = slice(0, d); dst[ax] = slice(-d, None)
                            sh[tuple(dst)] = nb[tuple(src)]
                            nb = sh
                    out &= nb
        m = out
    return m


@pytest.mark.parametrize("per", [(1, 1, 1), (0, 0, 0), (1, 0, 1), (0, 1, 0)])
@pytest.mark.parametrize("passes", [1, 2, 3])
def test_nesting_erosion_matches_a_plain_restatement(per, passes):
    lib, L = _lib()
    rng = np.random.default_rng(11 + passes)
    for case in range(12):
        n = tuple(int(v) for v in rng.integers(5, 20, 3))
        m = np.zeros(n, np.uint8)
        for _ in range(int(rng.integers(1, 5))):
            lo = [int(rng.integers(0, n[d])) for d in range(3)]
            hi = [int(rng.integers(lo[d], n[d])) for d in range(3)]
            for d in range(3):
                if rng.random() < 0.3: lo[d] = 0
                if rng.random() < 0.3: hi[d] = n[d] - 1
            m[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1] = 1
        ref = _erode_ref(m, per, passes)
        buf = np.ascontiguousarray(m.transpose(2, 1, 0))          # x fastest
        lib.check(L.iamrx_host_erode(buf.ctypes.data_as(C.POINTER(C.c_ubyte)), lib.i3(n), lib.i3(per), passes))
        assert np.array_equal(buf.transpose(2, 1, 0), ref), (n, per, passes, case)


def _cluster(lib, L, tags, bf, mgs, eff, nbuf, allowed=None):
    n = tags.shape
    t = np.ascontiguousarray(tags.transpose(2, 1, 0).astype(np.uint8))
    a = np.ascontiguousarray(allowed.transpose(2, 1, 0).astype(np.uint8)) if allowed is not None else None
    cap = 4096
    boxes = (C.c_int * (6 * cap))()
    nb = C.c_int(cap)
    lib.check(L.iamrx_host_cluster_tags(t.ctypes.data_as(C.POINTER(C.c_ubyte)), lib.i3((0, 0, 0)), lib.i3(tuple(v - 1 for v in n)), bf, mgs, C.c_double(eff), nbuf,
                                        a.ctypes.data_as(C.POINTER(C.c_ubyte)) if a is not None else None, boxes, C.byref(nb)))
    return [(tuple(boxes[6 * q + d] for d in range(3)), tuple(boxes[6 * q + 3 + d] for d in range(3))) for q in range(nb.value)]


def _blocks_ref(tags, bf, nbuf):
    """TagBoxArray::buffer(nbuf) clipped at the domain, then coarsen(bf): the blocks that hold a buffered tag"""
    n = tags.shape
    buf = np.zeros(n, bool)
    idx = np.argwhere(tags)
    for i, j, k in idx:
        buf[max(0, i - nbuf):i + nbuf + 1, max(0, j - nbuf):j + nbuf + 1, max(0, k - nbuf):k + nbuf + 1] = True
    nc = tuple(v // bf for v in n)
    return buf.reshape(nc[0], bf, nc[1], bf, nc[2], bf).any(axis=(1, 3, 5))


@pytest.mark.parametrize("bf,mgs,nbuf", [(4, 16, 1), (8, 32, 2), (2, 8, 0), (4, 8, 3)])
def test_cluster_boxes_cover_the_buffered_tags_and_nothing_forbidden(bf, mgs, nbuf):
    lib, L = _lib()
    rng = np.random.default_rng(5 + bf + nbuf)
    for case in range(6):
        n = tuple(int(v) * bf for v in rng.integers(3, 9, 3))
        X, Y, Z = np.meshgrid(*[(np.arange(v) + 0.5) / v for v in n], indexing="ij")
        c = rng.random(3)
        tags = (np.abs(np.sqrt((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) - 0.3) < 0.04) | (rng.random(n) < 0.002)
        use_allowed = case % 2 == 1
        allowed = None
        if use_allowed:
            allowed = np.ones(n, bool)
            allowed[: n[0] // 3] = False
            allowed[:, :, -bf:] = False
        bx = _cluster(lib, L, tags, bf, mgs, 0.7, nbuf, allowed)
        want = _blocks_ref(tags, bf, nbuf)
        if use_allowed:
            nc = want.shape
            okb = allowed.reshape(nc[0], bf, nc[1], bf, nc[2], bf).all(axis=(1, 3, 5))
            want &= okb
        cover = np.zeros(n, int)
        for lo, hi in bx:
            for d in range(3):
                assert lo[d] % bf == 0 and (hi[d] + 1) % bf == 0 and 0 <= lo[d] <= hi[d] < n[d] and hi[d] - lo[d] + 1 <= mgs
            cover[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1] += 1
        assert cover.max() <= 1, "boxes overlap"
        nc = want.shape
        covb = cover.reshape(nc[0], bf, nc[1], bf, nc[2], bf).all(axis=(1, 3, 5))
        assert np.all(covb[want]), "a block with a buffered tag is not covered"
        if use_allowed:
            assert not np.any(cover.astype(bool) & ~np.repeat(np.repeat(np.repeat(okb, bf, 0), bf, 1), bf, 2)), "a box covers a block that is not allowed"
        eff = want.sum() * bf ** 3 / max(1, cover.sum())
        assert eff > 0.3 or want.sum() == 0, eff
