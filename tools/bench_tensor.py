"""Fused tensor residual / apply at n^3 (periodic, constant viscosity): ms per launch of the face-flux form (TENSOR_UNI_CC = 0) and the
cell-centred form k_tensor_uni in its tile shapes (1: 32 x 8, 2: 32 x 16, 3: 64 x 8, 4: 64 x 4, 5: 16 x 16), and their largest difference.
  python tools/bench_tensor.py [n]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iamr_amd import lib, ns as N
lib.init(0)
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 256,) * 3
g = lib.Geom.make(n)
lay = lib.Layout.single(n)
rng = np.random.default_rng(3)
u = lib.MultiFab(lay, lib.CELL, 3, 1)
ua = rng.standard_normal(tuple(v + 2 for v in n) + (3,))
u.set_from_global(ua, (-1, -1, -1))
u.fill_boundary(g)
a_d = lib.MultiFab(lay, lib.CELL, 1, 0); a_d.set_from_global(1.0 + 0.2 * rng.random(n + (1,)), (0, 0, 0))
eta = []
for d in range(3):
    m = lib.MultiFab(lay, lib.face(d), 1, 0); m.setval(0.013); eta.append(m)
out = lib.MultiFab(lay, lib.CELL, 3, 0)
ref = None
for cc in (0, 1, 2, 3, 4, 5):
    lib.tuning_set("TENSOR_UNI_CC", cc)
    for (a, b, ac) in ((1.0, 0.004, a_d), (0.0, -1.0, None)):
        for _ in range(3): N.tensor_apply(g, out, u, a, b, ac, eta)
        lib.sync(); t0 = time.perf_counter()
        for _ in range(20): N.tensor_apply(g, out, u, a, b, ac, eta)
        lib.sync(); ms = (time.perf_counter() - t0) / 20 * 1e3
        got = out.gather_valid(n)
        if cc == 0 and ac is not None: ref = got
        d = float(np.abs(got - ref).max() / np.abs(ref).max()) if (ac is not None and ref is not None) else float("nan")
        print(f"cc={cc} a={a} : {ms:.3f} ms per apply (incl. ghost fills), rel diff vs face-flux form {d:.2e}", flush=True)
