"""k_nodal_gsr: time of one sweep (two launches, index wrap) against the number of z-chunks per tile (IAMRX_GSR_NPL) -- the fixed cost of a
workgroup (prologue: three x planes, two sigma planes, right-hand side; launch ramp) against the length of its march (scratch tool)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
from iamr_amd import lib
from iamr_amd import ns as N
lib.init(0)
L = lib.lib()
def ev(fn, reps):
    for _ in range(3): fn()
    lib.sync(); lib.check(L.iamrx_timer_start())
    for _ in range(reps): fn()
    ms = C.c_double(); lib.check(L.iamrx_timer_stop(C.byref(ms))); return ms.value / reps
for n in [int(a) for a in sys.argv[1:]] or [256]:
    g = lib.Geom.make((n,) * 3); lay = lib.Layout.single((n,) * 3)
    sig = lib.MultiFab(lay, lib.CELL, 1, 4); sig.setval(1.0)
    x = lib.MultiFab(lay, lib.NODE, 1, 4); r = lib.MultiFab(lay, lib.NODE, 1, 4); x.setval(0.2); r.setval(1.0)
    for slots in (256, 512):
        lib.tuning_set("GSR_SLOTS", slots)
        for npl in (1, 2, 3, 4, 5, 7, 10, 13, 20, 26, 43, 65, 129):
            lib.tuning_set("GSR_NPL", npl)
            t = ev(lambda: N.nodal_gs_sweep(g, x, r, sig, 4), 10)
            print(f"n {n} npl {npl:4d} us/launch {t * 500:8.1f}", flush=True)
        break
    lib.tuning_set("GSR_NPL", 0)
