"""per-level cost of one nodal GS sweep (scratch tool)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
from iamr_amd import lib
from iamr_amd import ns as N
lib.init(0)
L = lib.lib()
def ev(fn, reps):
    for _ in range(2): fn()
    lib.sync(); lib.check(L.iamrx_timer_start())
    for _ in range(reps): fn()
    ms = C.c_double(); lib.check(L.iamrx_timer_stop(C.byref(ms))); return ms.value/reps
for n in (256,128,64,32,16,8):
    g = lib.Geom.make((n,)*3); lay = lib.Layout.single((n,)*3)
    res = {}
    for mode, ng in ((0,1),(1,4),(2,1)):
        if mode==2 and n>32: continue
        sig = lib.MultiFab(lay, lib.CELL, 1, ng); sig.setval(1.0)
        x = lib.MultiFab(lay, lib.NODE, 1, ng); r = lib.MultiFab(lay, lib.NODE, 1, ng); x.setval(0.2); r.setval(1.0)
        res[mode] = ev(lambda: N.nodal_gs_sweep(g, x, r, sig, mode), 10)
        t0=time.perf_counter(); 
        for _ in range(10): N.nodal_gs_sweep(g, x, r, sig, mode)
        lib.sync(); res[(mode,'wall')] = (time.perf_counter()-t0)/10*1e3
    print(n, {k: round(v,4) for k,v in res.items()})
    # fill only
    x = lib.MultiFab(lay, lib.NODE, 1, 1)
    print("   fill ng1", round(ev(lambda: x.fill_boundary(g), 20),4))
