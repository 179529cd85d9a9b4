"""HIP path against the committed fixtures of tests/golden/ (oracle-generated regression pins, see make_golden.py)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_taylorgreen16_fixture(gpu):
    from iamr_amd import ns as N
    lib = gpu
    gold = np.load(os.path.join(HERE, "golden", "taylorgreen16_oracle.npz"))
    n = (16, 16, 16)
    ns = N.NavierStokes(lib.Geom.make(n), lib.Layout.single(n), N.ns_params(cfl=0.5, visc_coef=float(gold["visc"]), init_iter=2))
    ns.init_taylorgreen(1.0, 1.0, 1.0, float(gold["c"]), 1.0)
    ns.post_init(-1.0)
    for _ in range(int(gold["nsteps"])):
        ns.step()
    assert abs(ns.time - float(gold["time"])) <= 1e-12
    S = ns.data(N.NavierStokes.S_NEW).gather_valid(n)
    assert np.abs(S - gold["S"]).max() <= 1e-8


def test_liddrivencavity16_fixture(gpu):
    from iamr_amd import ns as N
    lib = gpu
    gold = np.load(os.path.join(HERE, "golden", "liddrivencavity16_oracle.npz"))
    n = (16, 16, 16)
    lid = [0.0] * 9
    lid[6] = 1.0
    par = N.ns_params(cfl=0.3, visc_coef=0.01, init_dt=0.0140625, init_shrink=0.3, init_iter=3, tracer_diff_coef=0.001,
                      phys_lo=[4, 4, 5], phys_hi=[5, 5, 5], wall_vel_hi=lid)
    ns = N.NavierStokes(lib.Geom.make(n, periodic=(0, 0, 0)), lib.Layout.decompose(n, 8), par)
    ns.init_rest(1.0)
    ns.post_init(-1.0)
    dts = [ns.step() for _ in range(int(gold["nsteps"]))]
    assert np.allclose(dts, gold["dts"], rtol=1e-9, atol=0)
    S = ns.data(N.NavierStokes.S_NEW).gather_valid(n)
    assert np.abs(S - gold["S"]).max() <= 1e-8


def test_liddrivencavity16_from_inputs_file(gpu):
    """the same run driven by an IAMR-format inputs file through iamr_amd.inputs / iamr_amd.run (row f4)"""
    from iamr_amd import ns as N
    from iamr_amd.inputs import Inputs
    from iamr_amd.run import build
    lib = gpu
    gold = np.load(os.path.join(HERE, "golden", "liddrivencavity16_oracle.npz"))
    inp = Inputs([os.path.join(HERE, "golden", "inputs.3d.lid_driven_cavity16")])
    ns, lay, g, pr = build(inp, lib, N)
    assert lay.nlocal() == 8 and pr["max_step"] == 5
    ns.post_init(pr["stop_time"])
    dts = [ns.step() for _ in range(pr["max_step"])]
    assert np.allclose(dts, gold["dts"], rtol=1e-9, atol=0)
    S = ns.data(N.NavierStokes.S_NEW).gather_valid((16, 16, 16))
    assert np.abs(S - gold["S"]).max() <= 1e-8


def test_plotfile_of_the_golden_run(gpu, tmp_path):
    """row f2: the state of the inputs-driven run written as an AMReX plotfile (iamr_amd.plotfile) and read back: the file holds
    the golden state of the oracle run box by box"""
    from iamr_amd import ns as N
    from iamr_amd.inputs import Inputs
    from iamr_amd.run import build, write_plot
    from iamr_amd.plotfile import PlotFile
    lib = gpu
    gold = np.load(os.path.join(HERE, "golden", "liddrivencavity16_oracle.npz"))
    inp = Inputs([os.path.join(HERE, "golden", "inputs.3d.lid_driven_cavity16")])
    ns, lay, g, pr = build(inp, lib, N)
    ns.post_init(pr["stop_time"])
    for _ in range(pr["max_step"]):
        ns.step()
    path = write_plot(ns, lay, pr, N, pr["max_step"], str(tmp_path / "plt"))
    assert path.endswith("plt00005")
    pf = PlotFile.read(path)
    assert pf.names == ["x_velocity", "y_velocity", "z_velocity", "density", "tracer"] and len(pf.levels) == 1
    assert pf.levels[0].domain == ((0, 0, 0), (15, 15, 15)) and len(pf.levels[0].boxes) == 8 and abs(pf.time - ns.time) < 1e-15
    for (lo, hi), a in zip(pf.levels[0].boxes, pf.levels[0].data):
        ref = gold["S"][lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1, :]
        assert np.abs(a - ref).max() <= 1e-8
