import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iamr_amd import lib
lib.init(0)
n = 256
g = lib.Geom.make((n,)*3); lay = lib.Layout.single((n,)*3)
vel = lib.MultiFab(lay, lib.CELL, 3, 3); frc = lib.MultiFab(lay, lib.CELL, 3, 1)
um = [lib.MultiFab(lay, lib.face(d), 1, 1) for d in range(3)]
vel.setval(0.3); frc.setval(0.1)
aofs = lib.MultiFab(lay, lib.CELL, 5, 0); divu = lib.MultiFab(lay, lib.CELL, 1, 1); divu.setval(0.0)
dt = 0.3/n
for _ in range(3):
    lib.godunov_extrap_vel_to_faces(g, vel, frc, um, dt)
    lib.godunov_compute_aofs(g, aofs, 0, vel, 3, frc, divu, um, (0,0,0), dt, None, 1, 0)
lib.sync()

