"""the bench's AMR workload on its own (2-level TaylorGreen, nu = 1e-4): wall time per coarse step and the sections it is made of
(scratch tool; `python tools/bench_amr.py [n0] [steps]`)"""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iamr_amd import lib
import bench
lib.init(0)
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 128
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
print(json.dumps(bench.amr_workload(lib, n0, steps)))
