import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from iamr_amd import lib
from iamr_amd import ns as N
lib.init(0)
n=int(sys.argv[1])
g = lib.Geom.make((n,)*3); lay = lib.Layout.single((n,)*3)
vel = lib.MultiFab(lay, lib.CELL, 3, 1)
rng=np.random.default_rng(0)
a = np.asfortranarray(rng.standard_normal((n+2,)*3+(3,)))
sig = lib.MultiFab(lay, lib.CELL, 1, 1); sig.setval(1.0)
def nodal(tag):
    vel.from_numpy(a); vel.fill_boundary(g)
    phi = lib.MultiFab(lay, lib.NODE, 1, 1); phi.setval(0.0)
    st = N.nodal_projection(g, vel, 0, phi, sig, rel_tol=1e-9, opts=lib.mg_opts())
    print(tag, st.iters, round(st.vcycle_ms,2))
nodal("first"); nodal("second"); nodal("third")
# a MAC-like cell solve in between
b = [lib.MultiFab(lay, lib.face(d), 1, 0) for d in range(3)]
for m in b: m.setval(1.0)
phi = lib.MultiFab(lay, lib.CELL, 1, 1); rhs = lib.MultiFab(lay, lib.CELL, 1, 0); phi.setval(0.0)
r = rng.standard_normal((n,n,n,1)); r -= r.mean(); rhs.from_numpy(np.asfortranarray(r))
st = lib.abec_solve(g, 0.0, 1.0, None, b, phi, rhs, rtol=1e-10)
print("abec", st.iters, round(st.vcycle_ms,2))
nodal("after abec"); nodal("again")
ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.7, visc_coef=1e-4))
ns.init_taylorgreen(1,1,1,1,1); ns.post_init(-1.0)
for i in range(2):
    ns.step(); sm, sn, sv = ns.stats(); print("ns step", sn.iters, round(sn.vcycle_ms,2), "mac", round(sm.vcycle_ms,2))
nodal("after ns")
