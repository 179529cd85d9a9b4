"""level-wide BLAS-1 style launches on a level of MANY small boxes against one box of the same size (scratch: what a Krylov iteration on
the coarsest multigrid level of a regridded refined level costs per launch; BASELINE config C5)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C
from iamr_amd import lib
lib.init(0)
L = lib.lib()
def ev(fn, reps=50):
    for _ in range(5): fn()
    lib.sync(); lib.check(L.iamrx_timer_start())
    for _ in range(reps): fn()
    ms = C.c_double(); lib.check(L.iamrx_timer_stop(C.byref(ms))); return ms.value / reps * 1e3
def layouts():
    # (name, domain n, boxes)
    out = []
    bs = [((16 * i, 16 * j, 16 * k), (16 * i + 15, 16 * j + 15, 16 * k + 15)) for k in range(3) for j in range(12) for i in range(12)]
    out.append(("432 boxes of 16^3 (1.77 M cells)", (192, 192, 48), bs))
    mixed = []
    for j in range(12):
        for i in range(0, 12, 4):
            if (i + j) % 3 == 0: mixed.append(((16 * i, 16 * j, 0), (16 * i + 63, 16 * j + 15, 47)))
            else: mixed += [((16 * (i + q), 16 * j, 16 * k), (16 * (i + q) + 15, 16 * j + 15, 16 * k + 15)) for q in range(4) for k in range(3)]
    out.append((f"{len(mixed)} mixed boxes 16^3 and 64x16x48 (1.77 M cells)", (192, 192, 48), mixed))
    out.append(("1 box 192x192x48 (1.77 M cells)", (192, 192, 48), [((0, 0, 0), (191, 191, 47))]))
    b8 = [((8 * i, 8 * j, 8 * k), (8 * i + 7, 8 * j + 7, 8 * k + 7)) for k in range(3) for j in range(12) for i in range(12)]
    out.append(("432 boxes of 8^3 (221 k cells)", (96, 96, 24), b8))
    tiny = [((4 * i, 4 * j, 4 * k), (4 * i + 3, 4 * j + 3, 4 * k + 3)) for k in range(3) for j in range(12) for i in range(12)]
    out.append(("432 boxes of 4^3 (27 k cells)", (48, 48, 12), tiny))
    return out
for name, n, boxes in layouts():
    g = lib.Geom.make(n)
    lay = lib.Layout(boxes)
    for typ, tn in ((lib.CELL, "cell"), (lib.NODE, "node")):
        a = lib.MultiFab(lay, typ, 1, 1); b = lib.MultiFab(lay, typ, 1, 1)
        a.setval(1.0); b.setval(2.0)
        t_copy = ev(lambda: lib.check(L.iamrx_mf_copy(a.h, b.h, 0, 0, 1, 0)))
        t_set = ev(lambda: a.setval(0.5))
        t_fb = ev(lambda: a.fill_boundary(g))
        out = C.c_double()
        t_norm = ev(lambda: lib.check(L.iamrx_mf_norm0(a.h, 0, 1, 0, C.byref(out))), 20)
        print(f"{name:55s} {tn}: copy {t_copy:7.1f} us  setval {t_set:7.1f} us  fill_boundary {t_fb:7.1f} us  norm0 (incl. read-back) {t_norm:7.1f} us", flush=True)
    # the nodal operator of a Krylov iteration on such a level: the z-marching LDS kernel (tiles of 32 x 8 nodes) against the plain per-node form
    x = lib.MultiFab(lay, lib.NODE, 1, 1); r = lib.MultiFab(lay, lib.NODE, 1, 0); sg = lib.MultiFab(lay, lib.CELL, 1, 1)
    x.setval(1.0); sg.setval(1.0)
    ts = []
    for zm in (1, 0):
        lib.tuning_set("NODAL_RES_ZM", zm)
        ts.append(ev(lambda: lib.check(L.iamrx_nodal_residual(C.byref(g), r.h, x.h, sg.h, None))))
    lib.tuning_set("NODAL_RES_ZM", 1)
    print(f"{name:55s} nodal residual: z-marching {ts[0]:7.1f} us  per-node {ts[1]:7.1f} us", flush=True)
