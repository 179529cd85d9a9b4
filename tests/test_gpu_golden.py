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
