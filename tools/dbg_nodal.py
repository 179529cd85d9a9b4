import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from iamr_amd import lib
from iamr_amd import ns as N
lib.init(0)
n=64
g = lib.Geom.make((n,)*3); lay = lib.Layout.single((n,)*3)
vel = lib.MultiFab(lay, lib.CELL, 3, 1)
rng=np.random.default_rng(0)
a = rng.standard_normal((n+2,)*3+(3,)); vel.from_numpy(np.asfortranarray(a)); vel.fill_boundary(g)
sig = lib.MultiFab(lay, lib.CELL, 1, 1); sig.setval(1.0)
phi = lib.MultiFab(lay, lib.NODE, 1, 1); phi.setval(0.0)
st = N.nodal_projection(g, vel, 0, phi, sig, rel_tol=1e-9, opts=lib.mg_opts(verbose=1))
print(st.iters, st.vcycle_ms)
