"""bottom-solver iteration statistics of the three solves of a TaylorGreen step (scratch tool)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iamr_amd import lib
from iamr_amd import ns as N
lib.init(0)
for nn in (32, 128):
    n = (nn,) * 3
    g = lib.Geom.make(n); lay = lib.Layout.single(n)
    ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.7, visc_coef=1e-4, init_iter=2))
    ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
    ns.post_init(-1.0)
    for _ in range(2):
        ns.step()
        sm, sn, sv = ns.stats()
        print(nn, "mac iters", sm.iters, "bottom", sm.bottom_iters_total, "| nodal", sn.iters, sn.bottom_iters_total, "| visc", sv.iters, sv.bottom_iters_total, "levels", sm.nlevels, sn.nlevels)
