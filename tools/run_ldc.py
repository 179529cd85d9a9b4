"""LidDrivenCavity (the reference's regtest.3d.lid_driven_cavity, BASELINE config C4) at n^3 on one GPU: ms per step and V-cycle times;
IAMRX_GSRB_RB_WALLS=0 selects the two colour passes on the levels with walls (scratch tool)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iamr_amd import lib, ns as NS, run as R
from iamr_amd.inputs import Inputs
lib.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
inp = Inputs([os.path.join(ROOT, "tests", "golden", "regtest.3d.lid_driven_cavity")],
             [f"amr.n_cell={n} {n} {n}", f"amr.max_grid_size={n}", "max_step=8", f"ns.init_dt={0.0140625 * 64 / n}"])
pr = inp.problem()
ns, lay, g, pr = R.build(inp, lib, NS, 1, pr)
ns.post_init(pr["stop_time"])
for _ in range(2): ns.step()
lib.sync(); t0 = time.perf_counter()
for _ in range(6): ns.step()
lib.sync(); ms = (time.perf_counter() - t0) / 6 * 1e3
sm, sn, sv = ns.stats()
print(f"LDC {n}^3: {ms:.2f} ms/step, {n**3 / ms / 1e3:.1f} M cells/s; iters {sm.iters} {sn.iters} {sv.iters}; vcycle ms {sm.vcycle_ms:.2f} {sn.vcycle_ms:.2f} {sv.vcycle_ms:.2f}", flush=True)
