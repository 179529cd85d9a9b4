"""time-to-solution of the cell-centred MLMG (MAC-projection operator: variable b, periodic, 256^3) under different cycle parameters (scratch tool)"""
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
x = (np.arange(N) + 0.5) / N
X, Y, Z = np.meshgrid(x, x, x, indexing="ij")
a = (np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) * np.cos(2 * np.pi * Z) + 0.3 * np.sin(6 * np.pi * X) * np.sin(4 * np.pi * Y) + 0.1 * np.cos(32 * np.pi * Z) * np.sin(16 * np.pi * X))[..., None]
a -= a.mean()
rhs.from_numpy(np.asfortranarray(a))
cfgs = []
for nu in ((2, 2), (1, 1), (2, 1), (1, 2), (3, 3), (3, 2)):
    for om in (1.0, 1.15, 1.3):
        cfgs.append((f"nu{nu[0]}{nu[1]} omega {om}", dict(nu1=nu[0], nu2=nu[1], omega=om)))
for label, kw in [("default", {})] + cfgs:
    phi.setval(0.0)
    lib.sync(); t0 = time.perf_counter()
    try:
        st = lib.abec_solve(g, 0.0, 1.0, None, b, phi, rhs, rtol=1e-11, atol=1e-16, opts=lib.mg_opts(**kw))
        lib.sync(); wall = (time.perf_counter() - t0) * 1e3
        print(f"{label:20s} iters {st.iters} vcycle_ms {st.vcycle_ms:.2f} wall {wall:.1f} bottom_its {st.bottom_iters_total} res {st.resnorm:.2e}", flush=True)
    except Exception as e:
        print(label, "ERR", e, flush=True)
