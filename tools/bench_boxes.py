"""step time of TaylorGreen 256^3 on one GPU for different box decompositions (scratch tool)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iamr_amd import lib
from iamr_amd import ns as N
lib.init(0)
n = (256,) * 3
for mg in (256, 128, 64):
    g = lib.Geom.make(n); lay = lib.Layout.decompose(n, mg)
    ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.7, visc_coef=1e-4, init_iter=2))
    ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
    ns.post_init(-1.0)
    ns.step(); lib.sync()
    t0 = time.perf_counter()
    for _ in range(3): ns.step()
    lib.sync()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    sm, sn, sv = ns.stats()
    print(f"max_grid {mg}: boxes {lay.nlocal()} ms/step {ms:.1f} vcycle mac {sm.vcycle_ms:.2f} nodal {sn.vcycle_ms:.2f} visc {sv.vcycle_ms:.2f} iters {sm.iters} {sn.iters} {sv.iters}")
    del ns
