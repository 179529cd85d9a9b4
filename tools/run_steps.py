"""Runs TaylorGreen 256^3 (bench settings): post_init + 2 warm-up steps, then 4 steps between two marker launches (k_fill on a 7^3\nMultiFab) -- the workload of tools/profile_step.sh, which turns the rocprofv3 kernel trace between the markers into per-step totals."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iamr_amd import lib, ns as N
lib.init(0)
n = os.environ.get('IAMRX_N', '256')                                  # IAMRX_N=256,512,1024 IAMRX_MAXGRID=256: the shard proxy (8 boxes of 256^3, 1 x 2 x 4)
n = tuple(int(v) for v in n.split(',')) if ',' in n else (int(n),) * 3
mg = os.environ.get('IAMRX_MAXGRID', str(max(n)))
mg = tuple(int(v) for v in mg.split(',')) if ',' in mg else int(mg)      # IAMRX_MAXGRID=256,128,64: boxes that span x, split in y and z
g = lib.Geom.make(n, prob_hi=tuple(v / n[0] for v in n)); lay = lib.Layout.decompose(n, mg)     # IAMRX_MAXGRID=128: the 8-box decomposition
s = N.NavierStokes(g, lay, N.ns_params(cfl=0.7, visc_coef=1e-4, init_iter=2, init_shrink=1.0), lib.mg_opts())
s.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
s.post_init(-1.0)
for _ in range(2): s.step()
lib.sync()
print("MARK_BEGIN", flush=True)
# marker kernel: a distinctive fill size
m = lib.MultiFab(lib.Layout.single((7, 7, 7)), lib.CELL, 1, 0)
m.setval(1.0); m.setval(1.0); m.setval(1.0); lib.sync()   # marker: three consecutive tiny fills
t0 = time.perf_counter()
for _ in range(4): s.step()
lib.sync()
print("ms/step", (time.perf_counter() - t0) * 250)
m.setval(2.0); m.setval(2.0); m.setval(2.0); lib.sync()
