"""config C3 (DoubleShearLayer 2D, n^2 base + refined level) through bench.py's workload: python tools/run_c3.py [n] [steps]"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from iamr_amd import lib
lib.init(0)
print(json.dumps(bench.c3_workload(lib, int(sys.argv[1]) if len(sys.argv) > 1 else 512, int(sys.argv[2]) if len(sys.argv) > 2 else 3)))
