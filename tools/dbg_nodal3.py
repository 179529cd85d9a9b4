import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iamr_amd import lib
from iamr_amd import ns as N
lib.init(0)
n=int(sys.argv[1])
g = lib.Geom.make((n,)*3); lay = lib.Layout.single((n,)*3)
ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.7, visc_coef=1e-4))
ns.init_taylorgreen(1,1,1,float(sys.argv[2]),1); ns.post_init(-1.0)
for i in range(2):
    ns.step(); sm, sn, sv = ns.stats()
    print("nodal iters", sn.iters, "vc", round(sn.vcycle_ms,2), "bottom", sn.bottom_iters_total, "| mac", sm.iters, round(sm.vcycle_ms,2), sm.bottom_iters_total, "| visc", sv.iters, round(sv.vcycle_ms,2), sv.bottom_iters_total)
