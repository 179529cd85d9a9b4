"""AMReX plotfile format pinned on the reference's own committed plotfile (tests/golden/plt0000_1: Header and
Level_0 of Exec/run2d/test_grids/plt0000_1, data files of the reference): the reader decodes it, the writer reproduces its Header
and Level_0/Cell_H text and the FAB bytes."""
import os
import numpy as np
from iamr_amd.plotfile import PlotFile, Level, compare, from_level_data

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "plt0000_1")


def test_reader_decodes_the_reference_plotfile():
    hdr = open(os.path.join(GOLD, "Header")).read().split("\n")
    assert hdr[0] == "NavierStokes-V1.1" and hdr[1] == "7"
    # level 0 only is committed: read it through the level reader
    lv = Level(((0, 0), (15, 15)), (0.0625, 0.0625), [])
    PlotFile._read_level(GOLD, "Level_0/Cell", lv, 7, True)
    assert lv.boxes == [((0, 0), (7, 7)), ((8, 0), (15, 7)), ((0, 8), (7, 15)), ((8, 8), (15, 15))]
    assert [a.shape for a in lv.data] == [(8, 8, 7)] * 4
    # the minima / maxima recorded in Cell_H are those of the decoded data
    assert abs(lv.data[0][..., 0].min() - (-0.000515868739177496)) < 1e-18
    assert abs(lv.data[3][..., 6].max() - 6.73684774808615) < 1e-13
    assert lv.data[0][..., 3].min() == 0.0 and lv.data[0][..., 3].max() == 1.0 and lv.data[0][..., 4].min() == 1.0   # tracer blob, temperature >= 1


def test_writer_reproduces_the_reference_text_and_bytes(tmp_path):
    names = ["x_velocity", "y_velocity", "density", "tracer", "temp", "divu", "dsdt"]
    lv0 = Level(((0, 0), (15, 15)), (0.0625, 0.0625), [])
    PlotFile._read_level(GOLD, "Level_0/Cell", lv0, 7, True)
    lv1 = Level(((0, 0), (31, 31)), (0.03125, 0.03125), [((8, 8), (15, 15)), ((16, 8), (23, 15))])
    lv2 = Level(((0, 0), (63, 63)), (0.015625, 0.015625), [((24, 24), (31, 27)), ((32, 24), (39, 27))])
    for lv in (lv1, lv2):
        lv.data = [np.zeros(tuple(h - q + 1 for q, h in zip(lo, hi)) + (7,)) for lo, hi in lv.boxes]
    pf = PlotFile(names, 0.0, (0.0, 0.0), (1.0, 1.0), [lv0, lv1, lv2], [2, 2], 0, "NavierStokes-V1.1")
    assert pf.header_text() == open(os.path.join(GOLD, "Header")).read()
    out = str(tmp_path / "plt")
    pf.write(out)
    ref_h = open(os.path.join(GOLD, "Level_0", "Cell_H")).read().replace("Cell_D_0000 ", "Cell_D_00000 ")
    got_h = open(os.path.join(out, "Level_0", "Cell_H")).read()
    # offsets differ by the width of the file name only if header lines differ: they do not
    assert got_h.split("\n")[:15] == ref_h.split("\n")[:15]
    g = [l for l in got_h.split("\n")[15:] if "," in l and not l[0].isdigit() or l.startswith("-") or l[:1].isdigit() and l.count(",") == 7]
    r = [l for l in ref_h.split("\n")[15:] if "," in l and not l[0].isdigit() or l.startswith("-") or l[:1].isdigit() and l.count(",") == 7]
    assert len(g) == len(r) == 8
    for a, b in zip(g, r):
        va, vb = [float(x) for x in a.split(",")[:-1]], [float(x) for x in b.split(",")[:-1]]
        assert np.allclose(va, vb, rtol=1e-14, atol=0)
    assert open(os.path.join(out, "Level_0", "Cell_D_00000"), "rb").read() == open(os.path.join(GOLD, "Level_0", "Cell_D_0000"), "rb").read()


def test_round_trip_and_compare(tmp_path):
    rng = np.random.default_rng(0)
    n = (8, 4, 6)
    boxes = [((0, 0, 0), (3, 3, 5)), ((4, 0, 0), (7, 3, 5))]
    arrs = [rng.standard_normal((4, 4, 6, 5)) for _ in boxes]
    pf = from_level_data(n, (0.0, 0.0, 0.0), (2.0, 1.0, 1.5), boxes, arrs, 0.125, 7)
    a, b = str(tmp_path / "a"), str(tmp_path / "b")
    pf.write(a)
    back = PlotFile.read(a)
    assert back.names == pf.names and back.time == 0.125 and back.levels[0].step == 7 and back.levels[0].boxes == boxes
    assert all(np.array_equal(x, y) for x, y in zip(back.levels[0].data, arrs))
    arrs2 = [x.copy() for x in arrs]
    arrs2[1][2, 1, 3, 4] += 1e-3
    from_level_data(n, (0.0, 0.0, 0.0), (2.0, 1.0, 1.5), boxes, arrs2, 0.125, 7).write(b)
    d = compare(a, b)
    assert d["tracer"][0] == 1e-3 or abs(d["tracer"][0] - 1e-3) < 1e-15
    assert all(d[k][0] == 0.0 for k in pf.names if k != "tracer")


def test_three_level_reference_plotfile_round_trip_and_grid_file(tmp_path):
    """tests/golden/plt0000_2 = Exec/run2d/test_grids/plt0000_2 (data files of the reference: three levels, ref_ratio 4 2): the reader decodes
    every level, the writer reproduces the Header text and the FAB bytes of every level, and the boxes of the refined levels are the
    ones of the run's fixed-grid file (tests/golden/fixed_grids_2 = Exec/run2d/test_grids/fixed_grids_2, amr.regrid_file of
    inputs_2_xy_ysolid) refined from the next coarser level's index space -- the convention iamr_amd.inputs.read_grid_file follows."""
    from iamr_amd.inputs import read_grid_file
    gold = os.path.join(os.path.dirname(GOLD), "plt0000_2")
    pf = PlotFile.read(gold)
    assert len(pf.levels) == 3 and pf.ref_ratio == [4, 2] and pf.names[:4] == ["x_velocity", "y_velocity", "density", "tracer"]
    assert pf.header_text() == open(os.path.join(gold, "Header")).read()
    grids = read_grid_file(os.path.join(os.path.dirname(GOLD), "fixed_grids_2"), [4, 2])
    assert len(grids) == 2
    for l in (1, 2):
        assert sorted(pf.levels[l].boxes) == sorted(grids[l - 1]), (l, pf.levels[l].boxes, grids[l - 1])
    out = str(tmp_path / "plt")
    pf.write(out)
    for l in range(3):
        assert open(os.path.join(out, f"Level_{l}", "Cell_D_00000"), "rb").read() == open(os.path.join(gold, f"Level_{l}", "Cell_D_0000"), "rb").read()
    back = PlotFile.read(out)
    assert all(np.array_equal(a, b) for l in range(3) for a, b in zip(back.levels[l].data, pf.levels[l].data))
