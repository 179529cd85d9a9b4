"""A/B of the nodal interpolation kernels (scratch tool): python tools/ab_interp.py OUT.npz [n]
Run once with IAMRX_NODAL_INTERP_LDS=0 and once with =1, then compare the two npz files bit for bit."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from iamr_amd import lib
from iamr_amd import ns as N
lib.init(0)
out = sys.argv[1]
res = {}
# ragged two-box case (box lengths not multiples of the tile, boxes of different size) + a single 8^3 level
for name, fboxes in (("ragged", [((0, 0, 0), (39, 11, 19)), ((40, 0, 0), (51, 11, 19))]), ("tiny", [((0, 0, 0), (7, 7, 7))])):
    cboxes = [(tuple(l // 2 for l in lo), tuple((h + 1) // 2 - 1 for h in hi)) for lo, hi in fboxes]
    fl, cl = lib.Layout(fboxes), lib.Layout(cboxes)
    fine, crse, sig = lib.MultiFab(fl, lib.NODE, 1, 1), lib.MultiFab(cl, lib.NODE, 1, 1), lib.MultiFab(fl, lib.CELL, 1, 1)
    rng = np.random.default_rng(5)
    for mf in (fine, crse, sig):
        for li in range(mf.nlocal()):
            a, lo = mf.to_numpy(li)
            mf.from_numpy(rng.random(a.shape) + 0.5, li)
    N.nodal_interp_add(fine, crse, sig)
    for li in range(fine.nlocal()):
        a, lo = fine.to_numpy(li)
        res[f"{name}{li}"] = a[1:-1, 1:-1, 1:-1].copy()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
for m in (n, n // 8):
    fl, cl = lib.Layout.single((m,) * 3), lib.Layout.single((m // 2,) * 3)
    fine, crse, sig = lib.MultiFab(fl, lib.NODE, 1, 1), lib.MultiFab(cl, lib.NODE, 1, 1), lib.MultiFab(fl, lib.CELL, 1, 1)
    fine.setval(0.0); crse.setval(1.0); sig.setval(1.0)
    N.nodal_interp_add(fine, crse, sig)
    lib.sync(); t0 = time.perf_counter()
    for _ in range(20):
        N.nodal_interp_add(fine, crse, sig)
    lib.sync()
    print(f"n={m}: {(time.perf_counter() - t0) / 20 * 1e6:.1f} us per call, env={os.environ.get('IAMRX_NODAL_INTERP_LDS')}")
    a, lo = fine.to_numpy(0)
    res[f"const{m}"] = np.array([a[1:-1, 1:-1, 1:-1].min(), a[1:-1, 1:-1, 1:-1].max()])
np.savez(out, **res)
