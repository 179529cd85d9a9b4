"""LidDrivenCavity n^3 (regtest.3d.lid_driven_cavity): 2 warm-up steps, then 4 steps between the marker launches of tools/profile_step.sh
(DBG=run_ldc_steps.py bash tools/profile_step.sh)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iamr_amd import lib, ns as NS, run as R
from iamr_amd.inputs import Inputs
lib.init(0)
n = int(os.environ.get("IAMRX_N", "256"))
inp = Inputs([os.path.join(ROOT, "tests", "golden", "regtest.3d.lid_driven_cavity")],
             [f"amr.n_cell={n} {n} {n}", f"amr.max_grid_size={n}", "max_step=8", f"ns.init_dt={0.0140625 * 64 / n}"])
pr = inp.problem()
ns, lay, g, pr = R.build(inp, lib, NS, 1, pr)
ns.post_init(pr["stop_time"])
for _ in range(2): ns.step()
lib.sync()
m = lib.MultiFab(lib.Layout.single((7, 7, 7)), lib.CELL, 1, 0)
m.setval(1.0); m.setval(1.0); m.setval(1.0); lib.sync()
t0 = time.perf_counter()
for _ in range(4): ns.step()
lib.sync()
print("ms/step", (time.perf_counter() - t0) * 250)
m.setval(2.0); m.setval(2.0); m.setval(2.0); lib.sync()
ns.profile(2)
for _ in range(2): ns.step()
sec = ns.profile(0)
print("sections", {k: round(v / 2, 2) for k, v in zip(["predict_velocity", "mac_project", "advection", "updates", "viscous", "nodal_project"], sec[:6])})
