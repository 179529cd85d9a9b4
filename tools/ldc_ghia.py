"""Lid-driven cavity Re = 100 run to steady state, centreline profile vs Ghia, Ghia & Shin (1982) (scratch / validation tool)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from iamr_amd import lib
from iamr_amd import ns as N
lib.init(0)
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
ny = int(sys.argv[3]) if len(sys.argv) > 3 else 4
n = (nx, ny, nx)
g = lib.Geom.make(n, prob_hi=(1.0, float(ny) / nx, 1.0), periodic=(0, 1, 0))
lay = lib.Layout.single(n)
lid = [0.0] * 9; lid[6] = 1.0
ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.8, visc_coef=0.01, init_dt=0.3 / nx, init_shrink=0.3, init_iter=3,
                                         phys_lo=[5, 0, 5], phys_hi=[5, 0, 5], wall_vel_hi=lid))
ns.init_rest(1.0)
ns.post_init(-1.0)
t0 = time.perf_counter(); steps = 0
while ns.time < T:
    ns.step(); steps += 1
lib.sync()
S = ns.data(N.NavierStokes.S_NEW).gather_valid(n)
u = S[..., 0]
z = (np.arange(nx) + 0.5) / nx
uc = 0.5 * (u[nx // 2 - 1, 0, :] + u[nx // 2, 0, :])          # x = 0.5
ghia_y = np.array([0.0547, 0.0625, 0.0703, 0.1016, 0.1719, 0.2813, 0.4531, 0.5, 0.6172, 0.7344, 0.8516, 0.9531, 0.9609, 0.9688, 0.9766])
ghia_u = np.array([-0.03717, -0.04192, -0.04775, -0.06434, -0.10150, -0.15662, -0.21090, -0.20581, -0.13641, 0.00332, 0.23151, 0.68717, 0.73722, 0.78871, 0.84123])
ui = np.interp(ghia_y, z, uc)
print("steps", steps, "time", ns.time, "wall s", time.perf_counter() - t0)
print("max |u - ghia|", np.abs(ui - ghia_u).max(), "min u", uc.min(), "y-uniformity", np.abs(u - u[:, :1, :]).max())
for a, b, c in zip(ghia_y, ghia_u, ui): print(f"{a:.4f} {b:+.5f} {c:+.5f}")
