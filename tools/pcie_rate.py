"""Host-buffer hand-over cost of one 256^3 level (scratch tool): iamrx_mf_from_host / iamrx_mf_to_host of the state (5 comps, 1 ghost),
pressure (nodes) and grad p (3 comps) -- what a caller that keeps its FABs in host memory would pay per step on top of the advance."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from iamr_amd import lib
lib.init(0)
n = 256
lay = lib.Layout.single((n,) * 3)
tot_b, tot_t = 0, 0.0
for name, typ, nc, up, down in (("state", lib.CELL, 5, True, True), ("pressure", lib.NODE, 1, False, True), ("gradp", lib.CELL, 3, False, True)):
    mf = lib.MultiFab(lay, typ, nc, 1)
    mf.setval(1.0)
    lo, hi = mf.fab_box(0)
    shape = tuple(hi[d] - lo[d] + 1 for d in range(3)) + (nc,)
    buf = np.ones(shape, order="F")
    p = buf.ctypes.data_as(C.POINTER(C.c_double))
    for direction, fn in (("H2D", lib.lib().iamrx_mf_from_host), ("D2H", lib.lib().iamrx_mf_to_host)):
        if (direction == "H2D" and not up) or (direction == "D2H" and not down):
            continue
        lib.check(fn(mf.h, 0, p)); lib.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            lib.check(fn(mf.h, 0, p))
        lib.sync()
        t = (time.perf_counter() - t0) / 3
        tot_b += buf.nbytes; tot_t += t
        print(f"{name:9s} {direction}: {buf.nbytes / 1e6:8.1f} MB in {t * 1e3:7.2f} ms = {buf.nbytes / t / 1e9:5.1f} GB/s")
print(f"per step (state up; state, pressure, grad p down): {tot_b / 1e9:.2f} GB, {tot_t * 1e3:.1f} ms")
