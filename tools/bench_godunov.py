"""ExtrapVelToFaces / ComputeAofs timings at n^3 (scratch tool)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
from iamr_amd import lib
lib.init(0)
L = lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
def ev(fn, reps):
    for _ in range(2): fn()
    lib.sync(); lib.check(L.iamrx_timer_start())
    for _ in range(reps): fn()
    ms = C.c_double(); lib.check(L.iamrx_timer_stop(C.byref(ms))); return ms.value / reps
g = lib.Geom.make((n,) * 3); lay = lib.Layout.single((n,) * 3)
vel = lib.MultiFab(lay, lib.CELL, 3, 3); frc = lib.MultiFab(lay, lib.CELL, 3, 1)
um = [lib.MultiFab(lay, lib.face(d), 1, 1) for d in range(3)]
vel.setval(0.3); frc.setval(0.1)
for m in um: m.setval(0.2)
aofs = lib.MultiFab(lay, lib.CELL, 5, 0); divu = lib.MultiFab(lay, lib.CELL, 1, 1); divu.setval(0.0)
dt = 0.3 / n
print("extrap", round(ev(lambda: lib.godunov_extrap_vel_to_faces(g, vel, frc, um, dt), 5), 3),
      "aofs_vel", round(ev(lambda: lib.godunov_compute_aofs(g, aofs, 0, vel, 3, frc, divu, um, (0, 0, 0), dt, None, 1, 0), 5), 3))
