"""steady-state memory check: live / cached bytes and hipMalloc count over many steps (scratch tool)"""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iamr_amd import lib
from iamr_amd import ns as N
lib.init(0)
n = (64, 64, 64)
g = lib.Geom.make(n); lay = lib.Layout.decompose(n, 32)
ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.7, visc_coef=1e-3, tracer_diff_coef=1e-3))
ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
ns.post_init(-1.0)
def info():
    live, cached, nm = C.c_size_t(), C.c_size_t(), C.c_size_t()
    lib.check(lib.lib().iamrx_mem_info(C.byref(live), C.byref(cached)))
    lib.check(lib.lib().iamrx_alloc_count(C.byref(nm)))
    return live.value, cached.value, nm.value
for it in range(41):
    ns.step()
    if it % 10 == 0:
        print(it, info())
