"""scratch: verbose solves of a chopped TaylorGreen level with the multi-box sweep kernel"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iamr_amd import lib, ns as N
lib.init(0)
lib.tuning_set("COALESCE", 0)
n, mg = (256, 32, 32), (128, 16, 16)
g = lib.Geom.make(n, prob_hi=(1.0, 0.125, 0.125)); lay = lib.Layout.decompose(n, mg)
s = N.NavierStokes(g, lay, N.ns_params(init_iter=2, cfl=0.7, visc_coef=1e-3), lib.mg_opts(verbose=1))
s.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
try:
    s.post_init(-1.0)
    s.step()
except Exception as e:
    print("EXC", e)
