"""the one-launch red + black sweep (k_abec_gsrb_rb) against the two colour passes (k_abec_gsrb2) at n^3 through iamrx_abec_form: MAC form
(coef 1: density read) and uniform coefficients (coef 2).  Run under `rocprofv3 --kernel-trace --stats` for per-kernel times (the entry's own
timings include its coefficient set-up) (scratch tool)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
from iamr_amd import lib
lib.init(0)
L = lib.lib()
def ev(fn, reps):
    for _ in range(3): fn()
    lib.sync(); lib.check(L.iamrx_timer_start())
    for _ in range(reps): fn()
    ms = C.c_double(); lib.check(L.iamrx_timer_stop(C.byref(ms))); return ms.value / reps
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
per = tuple(int(v) for v in os.environ.get("RB_PER", "1,1,1").split(","))          # RB_PER=0,0,0: a closed box (Neumann walls)
bc = tuple(0 if p else 102 for p in per)
g = lib.Geom.make((n,) * 3, periodic=per); lay = lib.Layout.single((n,) * 3)
rho = lib.MultiFab(lay, lib.CELL, 1, 1); rho.setval(1.0)
a = lib.MultiFab(lay, lib.CELL, 1, 1); b = lib.MultiFab(lay, lib.CELL, 1, 1); rhs = lib.MultiFab(lay, lib.CELL, 1, 0)
a.setval(0.5); b.setval(0.0); rhs.setval(1.0)
for coef in (1, 2):
    t = ev(lambda: (lib.abec_form(g, coef, 6, a, rhs, out=b, rho=rho, lobc=bc, hibc=bc), lib.abec_form(g, coef, 6, b, rhs, out=a, rho=rho, lobc=bc, hibc=bc)), 20) / 2
    print(f"n={n} coef={coef} one launch: {t*1e3:.1f} us per sweep (incl. the coefficient set-up of the entry)", flush=True)
    o0 = 4 if all(per) else 0
    t = ev(lambda: (lib.abec_form(g, coef, o0, a, rhs, rho=rho, lobc=bc, hibc=bc), lib.abec_form(g, coef, o0 + 1, a, rhs, rho=rho, lobc=bc, hibc=bc)), 20)
    print(f"n={n} coef={coef} two colour passes: {t*1e3:.1f} us per sweep", flush=True)
