"""config C3 (DoubleShearLayer 2D, n^2 base + refined level) through bench.py's workload: python tools/run_c3.py [n] [steps]
C3_SCOPES=1: scope profile (ProfScope: the stream is drained at both ends of every scope) of the whole run"""
import sys, os, json
import ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from iamr_amd import lib
lib.init(0)
if os.environ.get("C3_SCOPES"):
    lib.check(lib.lib().iamrx_scope_profile(1, 1, None, C.c_size_t(0)))
print(json.dumps(bench.c3_workload(lib, int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 3)))
if os.environ.get("C3_SCOPES"):
    buf = C.create_string_buffer(1 << 16)
    lib.check(lib.lib().iamrx_scope_profile(0, 0, buf, C.c_size_t(1 << 16)))
    print(buf.value.decode())
