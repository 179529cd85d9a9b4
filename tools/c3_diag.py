"""config C3 (bench.c3_workload's set-up): scoped timings of two coarse steps and their wall time (scratch diagnostic)"""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iamr_amd import lib, ns as NS, run as R
from iamr_amd.inputs import Inputs
lib.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
inp = Inputs([os.path.join(ROOT, "tests", "golden", "inputs.2d.doubleshearlayer_c3")], [f"amr.n_cell={n} {n}", "max_step=8", "proj.proj_tol=1.0e-10"])
pr = inp.problem()
amr, lays, g0 = R.build_amr(pr, lib, NS, 1)
amr.post_init(pr["stop_time"])
amr.coarse_step()
amr.coarse_step()
lib.sync()
print("grids", [len(l.boxes) for l in amr.layouts], "boxes l0", amr.layouts[0].boxes[:3], "l1", amr.layouts[1].boxes[:4] if amr.nlev > 1 else None)
L = lib.lib()
lib.check(L.iamrx_scope_profile(1, 1, None, C.c_size_t(0)))
m = lib.MultiFab(lib.Layout.single((7, 7, 7)), lib.CELL, 1, 0)
m.setval(1.0); m.setval(1.0); m.setval(1.0); lib.sync()
t0 = time.perf_counter()
for _ in range(4):
    amr.coarse_step()
lib.sync()
print("ms/step", (time.perf_counter() - t0) * 250)
m.setval(2.0); m.setval(2.0); m.setval(2.0); lib.sync()
buf = C.create_string_buffer(1 << 16)
lib.check(L.iamrx_scope_profile(0, 0, buf, C.c_size_t(1 << 16)))
print("scope profile of 4 coarse steps:")
print(buf.value.decode())
for l in range(amr.nlev):
    mm, nd, v = amr.levels[l].stats()
    print(f"level {l}: mac {mm.iters} ({mm.vcycle_ms:.3f} ms) nodal {nd.iters} ({nd.vcycle_ms:.3f}) visc {v.iters} ({v.vcycle_ms:.3f})")
