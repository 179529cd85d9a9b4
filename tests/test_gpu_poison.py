"""GPU: no kernel may read device memory that nobody wrote.  IAMRX_POISON_ALLOC=1 makes the library's caching allocator hand out every block
filled with 0xFF bytes (NaNs as doubles): a three-level subcycled viscous run (refined patches: coarse/fine ghost cells, Dirichlet masks,
registers, sync solves) must produce the same plotfile bytes with and without it."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _run(tmp_path, tag, poison):
    env = dict(os.environ)
    env["IAMRX_POISON_ALLOC"] = "1" if poison else "0"
    root = str(tmp_path / tag)
    cmd = [sys.executable, "-m", "iamr_amd.run", os.path.join(HERE, "golden", "inputs.3d.taylorgreen_amr16"), "ns.vel_visc_coef=0.01",
           "ns.scal_diff_coefs=0.005", f"amr.plot_file={root}"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return root + "00002"


def test_poisoned_allocator_gives_the_same_run(gpu, tmp_path):
    from iamr_amd.plotfile import PlotFile
    a = PlotFile.read(_run(tmp_path, "plain", False))
    b = PlotFile.read(_run(tmp_path, "poison", True))
    assert len(a.levels) == 3 and len(b.levels) == 3
    for la, lb in zip(a.levels, b.levels):
        assert la.boxes == lb.boxes
        for xa, xb in zip(la.data, lb.data):
            assert np.isfinite(xb).all()
            assert np.array_equal(xa, xb)
