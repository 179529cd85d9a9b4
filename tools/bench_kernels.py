"""Quick kernel timing on the GPU box (scratch tool; the judged numbers come from bench.py)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from iamr_amd import lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib.init(0)
n = (N, N, N)
g = lib.Geom.make(n)
lay = lib.Layout.single(n)
b = [lib.MultiFab(lay, lib.face(d), 1, 0) for d in range(3)]
for m in b:
    m.setval(1.0)
phi = lib.MultiFab(lay, lib.CELL, 1, 1)
rhs = lib.MultiFab(lay, lib.CELL, 1, 0)
out = lib.MultiFab(lay, lib.CELL, 1, 0)
phi.setval(0.0)
rng = np.random.default_rng(0)
a = rng.standard_normal((N, N, N, 1))
a -= a.mean()
rhs.from_numpy(np.asfortranarray(a))
cells = N ** 3


def timeit(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    lib.sync()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    lib.sync()
    return (time.perf_counter() - t) / reps


t = timeit(lambda: (lib.abec_gsrb(g, 0.0, 1.0, None, b, phi, rhs, 0), lib.abec_gsrb(g, 0.0, 1.0, None, b, phi, rhs, 1)))
print(f"GSRB red+black sweep {N}^3: {t*1e3:.3f} ms  -> {80*cells/t/1e9:.1f} GB/s algorithmic (80 B/cell)")
t = timeit(lambda: lib.abec_residual(g, 0.0, 1.0, None, b, out, phi, rhs))
print(f"residual {N}^3: {t*1e3:.3f} ms -> {48*cells/t/1e9:.1f} GB/s algorithmic (48 B/cell)")
t = timeit(lambda: phi.fill_boundary(g))
print(f"fill_boundary ng=1 {N}^3: {t*1e3:.3f} ms")
phi.setval(0.0)
st = lib.abec_solve(g, 0.0, 1.0, None, b, phi, rhs, rtol=1e-12, atol=1e-16, opts=lib.mg_opts(verbose=0))
print(f"MLMG solve {N}^3: iters {st.iters} levels {st.nlevels} vcycle {st.vcycle_ms:.3f} ms bottom_iters {st.bottom_iters_total} resnorm {st.resnorm:.3e}")
phi.setval(0.0)
st = lib.abec_solve(g, 0.0, 1.0, None, b, phi, rhs, rtol=1e-12, atol=1e-16, opts=lib.mg_opts(bottom_smoother_only=1))
print(f"MLMG solve (smoother bottom) {N}^3: iters {st.iters} vcycle {st.vcycle_ms:.3f} ms")
