"""scratch: the MAC solve of tests/test_gpu_rb_nbr.py::test_mac_solve_does_not_read... with verbose solver output"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from iamr_amd import lib
lib.init(0)
lib.tuning_set("COALESCE", 0)
PERIODIC, DIRICHLET, NEUMANN = 0, 101, 102
case = sys.argv[1] if len(sys.argv) > 1 else "channel"
per, lobc, hibc = {"channel": ((0, 0, 1), (NEUMANN, NEUMANN, PERIODIC), (DIRICHLET, NEUMANN, PERIODIC)),
                   "dir": ((0, 0, 0), (DIRICHLET,) * 3, (DIRICHLET,) * 3),
                   "wz": ((1, 1, 0), (PERIODIC, PERIODIC, NEUMANN), (PERIODIC, PERIODIC, NEUMANN))}[case]
smooth = len(sys.argv) > 2 and sys.argv[2] == "smooth"
n, mg = (256, 32, 32), (128, 16, 16)
g = lib.Geom.make(n, prob_hi=tuple(v / n[0] for v in n), periodic=per)
lay = lib.Layout.decompose(n, mg)
rng = np.random.default_rng(77)
if smooth:
    I, J, K = np.meshgrid(*[np.arange(-1, v + 1) for v in n], indexing="ij")
    rho = 1.0 + 0.3 * np.sin(2 * np.pi * I / n[0]) * np.cos(2 * np.pi * J / n[1]) * np.cos(2 * np.pi * K / n[2])
else:
    rho = 1.0 + 0.5 * rng.random(tuple(v + 2 for v in n))
for d in range(3):
    if per[d]:
        lo = [slice(None)] * 3; hi = [slice(None)] * 3; s0 = [slice(None)] * 3; s1 = [slice(None)] * 3
        lo[d] = 0; s0[d] = n[d]; hi[d] = n[d] + 1; s1[d] = 1
        rho[tuple(lo)] = rho[tuple(s0)]; rho[tuple(hi)] = rho[tuple(s1)]
um = []
for d in range(3):
    shp = tuple(n[e] + (1 if e == d else 0) for e in range(3))
    u = rng.standard_normal(shp)
    sl0 = [slice(None)] * 3; sl1 = [slice(None)] * 3; sl0[d] = 0; sl1[d] = n[d]
    if per[d]: u[tuple(sl1)] = u[tuple(sl0)]
    else:
        if lobc[d] == NEUMANN: u[tuple(sl0)] = 0.0
        if hibc[d] == NEUMANN: u[tuple(sl1)] = 0.0
    um.append(u)
for poison, nbr in ((False, 0), (False, 1), (True, 1)):
    lib.tuning_set("GSRB_RB_NBR", nbr)
    rho_d = lib.MultiFab(lay, lib.CELL, 1, 1)
    rho_d.set_from_global(rho[..., None], (-1,) * 3)
    um_d = []
    for d in range(3):
        m = lib.MultiFab(lay, lib.face(d), 1, 0); m.set_from_global(um[d][..., None], (0, 0, 0)); um_d.append(m)
    phi_d = lib.MultiFab(lay, lib.CELL, 1, 1); phi_d.setval(0.0)
    print("=== poison", poison, "nbr", nbr, flush=True)
    try:
        st = lib.mlmg_mac_solve(g, um_d, rho_d, 0, None, phi_d, 200.0, lobc=lobc, hibc=hibc, mac_tol=1e-10, opts=lib.mg_opts(maxorder=3, verbose=1, max_iters=12))
        print("iters", st.iters, "res", st.resnorm, "conv", st.converged, flush=True)
    except Exception as e:
        print("FAILED:", e, flush=True)
