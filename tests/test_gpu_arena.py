"""The device allocator (mf.hip Context::alloc, round 6: blocks carved out of large chunks, freed blocks merge with their free neighbours --
amrex::Arena's role).  What a user of MultiFabs can observe: arrays never overlap, the bytes in use return to where they were, and the space of
freed arrays serves LARGER arrays without another hipMalloc (the regrid of a growing level)."""
import ctypes as C
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mem(lib):
    live, cached = C.c_size_t(), C.c_size_t()
    lib.check(lib.lib().iamrx_mem_info(C.byref(live), C.byref(cached)))
    return live.value, cached.value


def _mallocs(lib):
    v = C.c_size_t()
    lib.check(lib.lib().iamrx_alloc_count(C.byref(v)))
    return v.value


def test_arrays_do_not_overlap_and_the_bytes_come_back(gpu):
    lib = gpu
    rng = np.random.default_rng(3)
    live0, _ = _mem(lib)
    alive = {}
    for it in range(200):
        if alive and rng.random() < 0.45:
            key = list(alive)[int(rng.integers(len(alive)))]
            mf, val = alive.pop(key)
            assert mf.norm0(0, mf.ncomp, mf.ngrow) == val        # nobody wrote into it while it lived
            del mf
        else:
            n = tuple(int(v) for v in rng.integers(3, 40, 3))
            nc, ng = int(rng.integers(1, 4)), int(rng.integers(0, 3))
            mf = lib.MultiFab(lib.Layout.single(n), lib.CELL if rng.random() < 0.5 else lib.NODE, nc, ng)
            val = float(it + 1)
            mf.setval(val)
            alive[it] = (mf, val)
    for mf, val in alive.values():
        assert mf.norm0(0, mf.ncomp, mf.ngrow) == val
    alive.clear()
    del mf
    import gc
    gc.collect()
    lib.sync()
    assert _mem(lib)[0] == live0


def test_freed_neighbours_merge_and_serve_a_larger_array(gpu):
    lib = gpu
    if lib.tuning_get("ARENA", 1) == 0:
        pytest.skip("IAMRX_ARENA = 0: blocks are cached by size, nothing merges")
    n = (48, 48, 48)
    warm = [lib.MultiFab(lib.Layout.single(n), lib.CELL, 1, 0) for _ in range(6)]     # six neighbours of 0.9 MB
    del warm
    m0 = _mallocs(lib)
    a = [lib.MultiFab(lib.Layout.single(n), lib.CELL, 1, 0) for _ in range(6)]
    del a
    big = lib.MultiFab(lib.Layout.single(n), lib.CELL, 5, 0)                           # 4.4 MB: only the merged extent holds it
    big.setval(2.0)
    assert big.norm0(0, 5, 0) == 2.0
    assert _mallocs(lib) == m0
