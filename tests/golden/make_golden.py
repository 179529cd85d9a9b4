"""Generates tests/golden/taylorgreen16_oracle.npz and tests/golden/liddrivencavity16_oracle.npz from the CPU oracle
(oracle/liborc.so).

The reference itself cannot be run here (AMReX / AMReX-Hydro absent), so this fixture is a
REGRESSION pin of the oracle (and of the HIP path through tests/test_gpu_golden.py), not a reference
golden vector: parity with upstream IAMR stays "unpinned" (see DESIGN.md).
Inputs: TaylorGreen 16^3, prob.c = 1, nu = 1e-2, cfl 0.5, init_iter 2; state after post_init + 3 steps.
        LidDrivenCavity 16^3 with the parameters of Exec/run3d/regtest.3d.lid_driven_cavity:5-46; state after post_init + 5 steps.
"""
import os
import sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc  # noqa: E402
from test_cpu_oracle import run_tg, run_ldc  # noqa: E402

if __name__ == "__main__":
    orc.build()
    S, P, t = run_tg(orc, 16, None, visc=1e-2, c=1.0, nsteps=3)
    np.savez_compressed(os.path.join(HERE, "taylorgreen16_oracle.npz"), S=S, P=P, time=t, visc=1e-2, c=1.0, nsteps=3)
    print("wrote fixture: time", t, "max|u|", np.abs(S[..., 0]).max())
    S, t, dts = run_ldc(orc, (16, 16, 16), (4, 4, 5), (5, 5, 5), 5)
    np.savez_compressed(os.path.join(HERE, "liddrivencavity16_oracle.npz"), S=S, time=t, dts=np.array(dts), nsteps=5)
    print("wrote LDC fixture: time", t, "max|u|", np.abs(S[..., 0]).max())
