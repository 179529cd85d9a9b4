"""the TaylorGreen 256^3 step of the bench under different multigrid cycle parameters (smoothing sweeps of the cell-centred and the nodal
cycles): ms per step and iterations / V-cycle time of the three solves (scratch tool)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iamr_amd import lib, ns as N
lib.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = lib.Geom.make((n, n, n)); lay = lib.Layout.single((n, n, n))
cfgs = [("default", {})]
for nu in ((3, 3), (2, 3), (3, 2), (4, 4), (1, 2), (1, 1)):
    cfgs.append((f"cell nu {nu[0]} {nu[1]}", dict(nu1=nu[0], nu2=nu[1])))
for nu in ((1, 1), (2, 1), (1, 2), (3, 3)):
    cfgs.append((f"nodal nu {nu[0]} {nu[1]}", dict(nodal_nu1=nu[0], nodal_nu2=nu[1])))
for label, kw in cfgs:
    s = N.NavierStokes(g, lay, N.ns_params(cfl=0.7, visc_coef=1e-4, init_iter=2, init_shrink=1.0), lib.mg_opts(**kw))
    s.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
    s.post_init(-1.0)
    for _ in range(2): s.step()
    lib.sync(); t0 = time.perf_counter()
    its = []
    for _ in range(6):
        s.step(); sm, sn, sv = s.stats(); its.append((sm.iters, sn.iters, sv.iters))
    lib.sync(); ms = (time.perf_counter() - t0) / 6 * 1e3
    print(f"{label:18s} {ms:7.2f} ms/step  iters (mac, nodal, visc) {its[-1]}  vcycle ms {sm.vcycle_ms:.2f} {sn.vcycle_ms:.2f} {sv.vcycle_ms:.2f}", flush=True)
    del s
