import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from iamr_amd import lib
lib.init(0)
import test_gpu_c3 as T
from iamr_amd.ns import ns_params
from iamr_amd.amr import Amr
n, nz, vort, mgs, bf = [int(v) for v in sys.argv[1:6]]
kw = dict(cfl=0.5, visc_coef=0.0, init_iter=2, init_shrink=1.0, do_cons_trac=1)
hz = 0.5 * nz * 2.0 / n
prob_lo, prob_hi = (-1.0, -1.0, -hz), (1.0, 1.0, hz)
g0 = lib.Geom.make([n, n, nz], prob_lo=prob_lo, prob_hi=prob_hi, periodic=(1, 1, 1))
lay0 = lib.Layout.decompose([n, n, nz], max(n, nz))
amr = Amr(g0, [lay0], ns_params(**kw), lib.mg_opts(verbose=1))
T._set_level(lib, amr.levels[0], lay0, [n, n, nz], prob_lo, prob_hi)
amr.set_regrid(max_level=1, regrid_int=1, rules=[dict(comp=-1, mode=2, value=[float(vort)])], blocking_factor=bf, max_grid_size=mgs, n_error_buf=1, grid_eff=0.75)
print(amr.regrid(), amr.nlev)
print(amr.layouts[1].boxes)
T._set_level(lib, amr.levels[1], amr.layouts[1], [2 * n, 2 * n, 2 * nz], prob_lo, prob_hi)
amr.post_init()
print("post_init ok")
amr.coarse_step()
print("step ok")
