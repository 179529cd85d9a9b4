"""BASELINE config C5 (RayleighTaylor, 3 levels on a 256^3 base) as bench.py runs it, alone: the bench's `rayleigh_taylor` object + the wall
time of its phases (scratch / profiling driver:  DBG=run_rt.py bash tools/profile_step.sh  uses the marker launches)"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iamr_amd import lib
import bench
lib.init(0)
n = int(os.environ.get("RT_N", "256"))
t0 = time.perf_counter()
kw = {}
if os.environ.get("RT_BSO"): kw["bottom_smoother_only"] = int(os.environ["RT_BSO"])
if os.environ.get("RT_NUF"): kw["nuf"] = int(os.environ["RT_NUF"])
if os.environ.get("RT_BRTOL"): kw["bottom_reltol"] = float(os.environ["RT_BRTOL"])
import ctypes as C
if os.environ.get("RT_SCOPES"):
    # scope profile (ProfScope: stream drained at both ends of every scope) of everything after post_init
    _orig = bench.time.perf_counter
    lib.check(lib.lib().iamrx_scope_profile(1, 1, None, C.c_size_t(0)))
res = bench.rt_workload(lib, n, int(os.environ.get("RT_STEPS", "2")), **kw)
if os.environ.get("RT_SCOPES"):
    buf = C.create_string_buffer(1 << 16)
    lib.check(lib.lib().iamrx_scope_profile(0, 0, buf, C.c_size_t(1 << 16)))
    print("scope profile (post_init + all coarse steps):")
    print(buf.value.decode())
print(json.dumps(res, indent=1))
print("total wall s", time.perf_counter() - t0)
