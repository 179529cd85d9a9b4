"""nodal residual and interpolation at n^3 nodes + 1 (periodic box, variable sigma) through the C-ABI; run under rocprofv3 --kernel-trace --stats for
the per-kernel times.  NODAL_RES_TILE / NODAL_INTERP_TILE variants are looped over.  python tools/bench_nodal_ops.py [n]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iamr_amd import lib, ns as N
lib.init(0)
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 256,) * 3
nc = tuple(v // 2 for v in n)
g = lib.Geom.make(n)
lay, clay = lib.Layout.single(n), lib.Layout.single(nc)
rng = np.random.default_rng(5)
sig = lib.MultiFab(lay, lib.CELL, 1, 1)
sig.set_from_global(1.0 + 0.3 * rng.random(tuple(v + 2 for v in n) + (1,)), (-1, -1, -1))
x = lib.MultiFab(lay, lib.NODE, 1, 1); r = lib.MultiFab(lay, lib.NODE, 1, 0); out = lib.MultiFab(lay, lib.NODE, 1, 0)
x.set_from_global(rng.standard_normal(tuple(v + 3 for v in n) + (1,)), (-1, -1, -1))
r.set_from_global(rng.standard_normal(tuple(v + 1 for v in n) + (1,)), (0, 0, 0))
c = lib.MultiFab(clay, lib.NODE, 1, 0)
c.set_from_global(rng.standard_normal(tuple(v + 1 for v in nc) + (1,)), (0, 0, 0))
ref = None
for tile in (0, 1, 2, 3):
    lib.tuning_set("NODAL_RES_TILE", tile)
    for _ in range(10): N.nodal_residual(g, out, x, sig, r)
    lib.sync()
    got = out.gather_valid(tuple(v + 1 for v in n))
    if ref is None: ref = got
    print("residual tile", tile, "identical to tile 0:", bool(np.array_equal(got, ref)), flush=True)
f0 = None
for tile in (0, 1):
    lib.tuning_set("NODAL_INTERP_TILE", tile)
    f = lib.MultiFab(lay, lib.NODE, 1, 1); f.setval(0.0)
    for _ in range(10): N.nodal_interp_add(f, c, sig)
    lib.sync()
    got = f.gather_valid(tuple(v + 1 for v in n))
    if f0 is None: f0 = got
    print("interp tile", tile, "identical:", bool(np.array_equal(got, f0)), flush=True)
