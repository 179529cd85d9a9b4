"""Launches the roofline-graded kernels a few times each, plus two calibration kernels with a known byte count (fill:
write-only, Copy: read+write of one 256^3 double array), for the rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE):
  cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -o f -- python tools/pmc_kernels.py
  (second pass with --pmc WRITE_SIZE);  tools/pmc_report.py turns the two counter CSVs into bytes per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iamr_amd import lib
from iamr_amd import ns as N
lib.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = lib.Geom.make((n,) * 3); lay = lib.Layout.single((n,) * 3)
a = lib.MultiFab(lay, lib.CELL, 1, 0); b = lib.MultiFab(lay, lib.CELL, 1, 0)
for _ in range(3):
    a.setval(1.0)                      # k_fill: n^3*8 bytes written
for _ in range(3):
    lib.check(lib.lib().iamrx_mf_copy(b.h, a.h, 0, 0, 1, 0)) if hasattr(lib.lib(), "iamrx_mf_copy") else None
bb = [lib.MultiFab(lay, lib.face(d), 1, 0) for d in range(3)]
for m in bb: m.setval(1.0)
phi = lib.MultiFab(lay, lib.CELL, 1, 1); rhs = lib.MultiFab(lay, lib.CELL, 1, 0); res = lib.MultiFab(lay, lib.CELL, 1, 0)
phi.setval(0.5); rhs.setval(1.0)
for _ in range(3):
    lib.abec_gsrb(g, 0.0, 1.0, None, bb, phi, rhs, 0); lib.abec_gsrb(g, 0.0, 1.0, None, bb, phi, rhs, 1)
    lib.abec_residual(g, 0.0, 1.0, None, bb, res, phi, rhs)
sig = lib.MultiFab(lay, lib.CELL, 1, 4); sig.setval(1.0)
x = lib.MultiFab(lay, lib.NODE, 1, 4); r = lib.MultiFab(lay, lib.NODE, 1, 4); o = lib.MultiFab(lay, lib.NODE, 1, 0)
x.setval(0.25); r.setval(1.0)
for _ in range(3):
    N.nodal_gs_sweep(g, x, r, sig, 1)
    N.nodal_residual(g, o, x, sig, r)
vel = lib.MultiFab(lay, lib.CELL, 3, 3); frc = lib.MultiFab(lay, lib.CELL, 3, 1)
um = [lib.MultiFab(lay, lib.face(d), 1, 1) for d in range(3)]
vel.setval(0.3); frc.setval(0.1)
for m in um: m.setval(0.2)
aofs = lib.MultiFab(lay, lib.CELL, 5, 0); divu = lib.MultiFab(lay, lib.CELL, 1, 1); divu.setval(0.0)
for _ in range(2):
    lib.godunov_extrap_vel_to_faces(g, vel, frc, um, 0.3 / n)
    lib.godunov_compute_aofs(g, aofs, 0, vel, 3, frc, divu, um, (0, 0, 0), 0.3 / n, None, 1, 0)
lib.sync()
print("done")
