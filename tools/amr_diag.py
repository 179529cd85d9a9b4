"""solver iteration counts of the bench's 2-level AMR workload, level by level (scratch diagnostic)"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iamr_amd import lib, ns as N
from iamr_amd.amr import Amr
lib.init(0)
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 128
g0 = lib.Geom.make((n0,) * 3)
lo, hi = n0 // 2, n0 // 2 + n0 - 1
lays = [lib.Layout.single((n0,) * 3), lib.Layout([((lo,) * 3, (hi,) * 3)])]
amr = Amr(g0, lays, N.ns_params(cfl=0.7, visc_coef=1e-4, init_iter=2, init_shrink=1.0), lib.mg_opts())
for l in range(2):
    amr.levels[l].init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
amr.post_init()
for step in range(3):
    amr.coarse_step()
    for l in range(2):
        m, nd, v = amr.levels[l].stats()
        print(f"step {step} level {l}: mac iters {m.iters} (vcycle {m.vcycle_ms:.3f} ms, res0 {m.resnorm0:.3e} rhs {m.rhsnorm0:.3e} res {m.resnorm:.3e}, levels {m.nlevels}) "
              f"nodal {nd.iters} ({nd.vcycle_ms:.3f} ms, levels {nd.nlevels}) visc {v.iters} ({v.vcycle_ms:.3f} ms)")
    s, ms = amr.sync_stats()
    print(f"   sync project {s.iters} cycles ({s.vcycle_ms:.3f} ms) res0 {s.resnorm0:.3e} res {s.resnorm:.3e}; mac_sync {ms.iters} ({ms.vcycle_ms:.3f} ms)")
import ctypes as C
L = lib.lib()
lib.check(L.iamrx_scope_profile(1, 1, None, C.c_size_t(0)))
for _ in range(2):
    amr.coarse_step()
buf = C.create_string_buffer(1 << 16)
lib.check(L.iamrx_scope_profile(0, 0, buf, C.c_size_t(1 << 16)))
print("scope profile of 2 coarse steps:")
print(buf.value.decode())
