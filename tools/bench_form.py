"""one colour pass of the pair-marching GSRB kernel (k_abec_gsrb2) at n^3 through iamrx_abec_form: MAC form (coef 1: density read) and uniform
coefficients (coef 2), index wrap and ghost-cell forms, for several planes-per-thread settings (scratch tool)"""
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
g = lib.Geom.make((n,) * 3); lay = lib.Layout.single((n,) * 3)
rho = lib.MultiFab(lay, lib.CELL, 1, 1); rho.setval(1.0)
phi = lib.MultiFab(lay, lib.CELL, 1, 1); rhs = lib.MultiFab(lay, lib.CELL, 1, 0)
phi.setval(0.5); rhs.setval(1.0)
for tz in [int(v) for v in os.environ.get("TZS", "32,16,8").split(",")]:
    lib.tuning_set("GSRB2_TZ", tz)
    for coef in (1, 2):
        for op, nm in ((4, "wrap"), (0, "ghost")):
            t = ev(lambda: (lib.abec_form(g, coef, op, phi, rhs, rho=rho), lib.abec_form(g, coef, op + 1, phi, rhs, rho=rho)), 20) / 2
            print(f"n={n} tz={tz} coef={coef} {nm}: {t*1e3:.1f} us per colour pass (incl. the coefficient set-up of the entry)", flush=True)
