"""GSRB sweep: two colour passes vs fused kernel (scratch tool)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
from iamr_amd import lib
lib.init(0)
L = lib.lib()
def ev(fn, reps):
    for _ in range(2): fn()
    lib.sync(); lib.check(L.iamrx_timer_start())
    for _ in range(reps): fn()
    ms = C.c_double(); lib.check(L.iamrx_timer_stop(C.byref(ms))); return ms.value / reps
for n in (256, 128, 64):
    g = lib.Geom.make((n,) * 3); lay = lib.Layout.single((n,) * 3)
    b = [lib.MultiFab(lay, lib.face(d), 1, 0) for d in range(3)]
    for m in b: m.setval(1.0)
    phi = lib.MultiFab(lay, lib.CELL, 1, 1); rhs = lib.MultiFab(lay, lib.CELL, 1, 0)
    phi.setval(0.5); rhs.setval(1.0)
    r = {}
    for fused in (0, 1):
        r[fused] = ev(lambda: lib.abec_gsrb_sweep(g, 0.0, 1.0, None, b, phi, rhs, 1.15, (0, 0, 0), (0, 0, 0), 2, fused), 10)
    print(n, {k: round(v, 4) for k, v in r.items()})
