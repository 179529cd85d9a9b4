"""nodal Gauss-Seidel sweep at n^3: the LDS-staged k_nodal_gs4 against the register-resident k_nodal_gsr (two launches = one sweep, no
ghost fills), variable and constant sigma, with and without index wrap (scratch tool)"""
import sys, os, json
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
sizes = [int(a) for a in sys.argv[1:]] or [256, 128, 64]
out = {}
for n in sizes:
    g = lib.Geom.make((n,) * 3); lay = lib.Layout.single((n,) * 3)
    import numpy as np
    rng = np.random.default_rng(1)
    sig = lib.MultiFab(lay, lib.CELL, 1, 4); sig.setval(1.0)
    x = lib.MultiFab(lay, lib.NODE, 1, 4); r = lib.MultiFab(lay, lib.NODE, 1, 4); x.setval(0.2); r.setval(1.0)
    nodes = (n + 1) ** 3
    csig = float(os.environ.get("GSR_BENCH_CSIG", "0"))
    lib.tuning_set("BENCH_CSIG", csig)
    for mode, mname in ((3, "fills_off"), (4, "wrap")):
        for tag, kv in (("gs4", dict(GSR=0)), ("gsr_pb4", dict(GSR=1, GSR_PB=4)), ("gsr_pb8", dict(GSR=1, GSR_PB=8))):
            for k, v in kv.items(): lib.tuning_set(k, v)
            t = ev(lambda: N.nodal_gs_sweep(g, x, r, sig, mode), 20)
            out[f"{n}_{mname}_{tag}"] = {"ms_per_sweep": round(t, 4), "alg_TBps": round(32 * nodes / t / 1e9, 3)}
            print(n, mname, tag, out[f"{n}_{mname}_{tag}"], flush=True)
    lib.tuning_set("GSR", 1); lib.tuning_set("GSR_PB", 4)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "bench_gsr.json"), "w"), indent=1)
