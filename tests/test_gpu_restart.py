"""checkpoint / restart (Amr::checkPoint / restart as IAMR drives them: NavierStokesBase::checkPoint / restart,
Source/NavierStokesBase.cpp:856-897, 2684-2727).  The procedure of the reference's own restart regression test
(Test/IAMR-tests.ini [Euler_restart]: Exec/run3d/regtest.3d.euler-restart, restartFileNum = 6): run to max_step = 10 writing a checkpoint
at step 6 (amr.check_int = 6), restart from chk00006, run to step 10 again and compare the final plotfiles -- here the comparison
demands EQUALITY TO THE BIT, which is what upstream's test harness checks with its zero tolerance."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _same_plotfiles(a, b):
    from iamr_amd.plotfile import PlotFile
    A, B = PlotFile.read(a), PlotFile.read(b)
    assert len(A.levels) == len(B.levels) and A.time == B.time
    for la, lb in zip(A.levels, B.levels):
        assert la.boxes == lb.boxes and la.step == lb.step
        for x, y in zip(la.data, lb.data):
            assert np.array_equal(x, y), float(np.abs(x - y).max())


def test_reference_euler_restart_regtest(gpu, tmp_path, capsys):
    """the reference's regtest.3d.euler-restart, unmodified (32^3 base, one refined level following the vorticity, regridded every
    second step, amr.check_int = 6): restarted run == uninterrupted run, bit for bit, grids included"""
    from iamr_amd import run as R
    inp = os.path.join(HERE, "golden", "regtest.3d.euler-restart")
    plt, chk = str(tmp_path / "plt"), str(tmp_path / "chk")
    assert R.main([inp, "amr.plot_int=10", f"amr.plot_file={plt}", f"amr.check_file={chk}"]) == 0
    out = capsys.readouterr().out
    assert "CHECKPOINT: " + chk + "00006" in out and len([l for l in out.splitlines() if l.startswith("STEP =")]) == 10
    hdr = open(chk + "00006/Header").read().split("\n")
    assert hdr[0] == "CheckPointVersion_1.0" and hdr[1] == "3" and hdr[3] == "1"          # max_level 1
    plt2 = str(tmp_path / "rst")
    assert R.main([inp, "amr.plot_int=10", f"amr.plot_file={plt2}", f"amr.check_file={chk}_b", f"amr.restart={chk}00006"]) == 0
    out = capsys.readouterr().out
    steps = [l for l in out.splitlines() if l.startswith("STEP =")]
    assert "RESTART from" in out and len(steps) == 4 and steps[0].startswith("STEP = 7 ") and steps[-1].startswith("STEP = 10 ")
    _same_plotfiles(plt + "00010", plt2 + "00010")


def test_single_level_viscous_restart(gpu, tmp_path, capsys):
    """one level, viscous (the Crank-Nicolson solves and the warm-started projections carry state from step to step): TaylorGreen inputs,
    checkpoint at step 3 of 6 (stop_time raised so that the step count, not the time, ends the run)"""
    from iamr_amd import run as R
    inp = os.path.join(HERE, "golden", "inputs.3d.taylorgreen")
    plt, chk = str(tmp_path / "plt"), str(tmp_path / "chk")
    args = [inp, "amr.n_cell=32 32 32", "max_step=6", "stop_time=100.0", "amr.plot_int=6", "amr.max_grid_size=16"]
    assert R.main(args + [f"amr.plot_file={plt}", f"amr.check_file={chk}", "amr.check_int=3"]) == 0
    capsys.readouterr()
    plt2 = str(tmp_path / "rst")
    assert R.main(args + [f"amr.plot_file={plt2}", "amr.check_int=-1", f"amr.restart={chk}00003"]) == 0
    out = capsys.readouterr().out
    assert len([l for l in out.splitlines() if l.startswith("STEP =")]) == 3
    _same_plotfiles(plt + "00006", plt2 + "00006")


def test_restart_with_a_later_stop_time(gpu, tmp_path, capsys):
    """a restart takes stop_time from the inputs file, as Amr::restart does, not from the checkpoint -- restarting with an
    extended stop_time is the usual reason to restart.  Run C stops at stop_time = 0.1 (three steps of ~0.02, no step clipped yet) and
    writes chk00003 holding that stop time; restarted with stop_time = 100 it must continue like the uninterrupted run A (stop_time = 100)
    through steps 4-6 -- with the checkpoint's stop time the level would clip step 5 to 0.1 and then advance with dt = 0."""
    from iamr_amd import run as R
    inp = os.path.join(HERE, "golden", "inputs.3d.taylorgreen")
    common = [inp, "amr.n_cell=32 32 32", "amr.max_grid_size=16"]
    pltA, pltD, chk = str(tmp_path / "pltA"), str(tmp_path / "pltD"), str(tmp_path / "chk")
    assert R.main(common + ["max_step=6", "stop_time=100.0", "amr.plot_int=6", f"amr.plot_file={pltA}", "amr.check_int=-1"]) == 0
    assert R.main(common + ["max_step=3", "stop_time=0.1", "amr.plot_int=-1", f"amr.check_file={chk}", "amr.check_int=3"]) == 0
    out = capsys.readouterr().out
    times = [float(l.split("TIME =")[1].split()[0]) for l in out.splitlines() if l.startswith("STEP =")]
    assert len(times) == 9 and times[-1] < 0.08, times                     # run C: three full steps, the fourth would still fit
    assert R.main(common + ["max_step=6", "stop_time=100.0", "amr.plot_int=6", f"amr.plot_file={pltD}", "amr.check_int=-1", f"amr.restart={chk}00003"]) == 0
    out = capsys.readouterr().out
    steps = [l for l in out.splitlines() if l.startswith("STEP =")]
    assert len(steps) == 3 and all(float(l.split("DT =")[1].split()[0]) > 0.01 for l in steps), steps
    _same_plotfiles(pltA + "00006", pltD + "00006")


def test_three_level_restart_with_regrids_above_level_zero(gpu, tmp_path, capsys):
    """Amr::level_count of EVERY level is checkpointed (max_level + 1 entries), not only level 0's: with amr.max_level = 2
    and regrid_int = 2 level 1 regrids level 2 inside the coarse steps, and when it does depends on its own count.  Checkpoint after an odd
    number of coarse steps; the restarted run reproduces the uninterrupted one bit for bit, grids included."""
    from iamr_amd import run as R
    from iamr_amd import checkpoint
    inp = os.path.join(HERE, "golden", "inputs.3d.tracer_regrid16")
    plt, chk = str(tmp_path / "plt"), str(tmp_path / "chk")
    assert R.main([inp, "max_step=6", "amr.plot_int=6", f"amr.plot_file={plt}", f"amr.check_file={chk}", "amr.check_int=3"]) == 0
    out = capsys.readouterr().out
    assert "CHECKPOINT: " + chk + "00003" in out
    hd = checkpoint.read_header(chk + "00003")
    assert hd["max_level"] == 2 and len(hd["level_count"]) == 3 and len(hd["dt_level"]) == 3 and len(hd["n_cycle"]) == 3
    plt2 = str(tmp_path / "rst")
    assert R.main([inp, "max_step=6", "amr.plot_int=6", f"amr.plot_file={plt2}", "amr.check_int=-1", f"amr.restart={chk}00003"]) == 0
    out = capsys.readouterr().out
    assert len([l for l in out.splitlines() if l.startswith("STEP =")]) == 3
    _same_plotfiles(plt + "00006", plt2 + "00006")
