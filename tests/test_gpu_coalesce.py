"""GPU: the level objects work on the caller's boxes MERGED (mf.h: coalesce_layout -- boxes of one owner that share full faces become one box; the
library's default, IAMRX_COALESCE = 1), which makes the time step independent of how amr.max_grid_size chopped the level
(reference Docs/sphinx_documentation/source/RunningProblems.rst:362-368: 32 by default): a level chopped into 64 boxes runs the single-box
kernels (index wrap on periodic domains, no ghost fills between colour passes) and gives the single-box answer TO THE BIT, while every data
accessor keeps speaking the caller's boxes.  The rest of the suite runs every test whose level objects merge boxes in both modes
(tests/conftest.py)."""
import ctypes as C
import os
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.merged_only]      # the tests below switch the mode themselves
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture()
def merged(gpu):
    before = gpu.tuning_get("COALESCE", 1)        # the suite's mode (0, or 1 under IAMRX_TEST_COALESCE=1): restored afterwards
    gpu.tuning_set("COALESCE", 1)
    yield gpu
    gpu.tuning_set("COALESCE", before)


def coalesced(lib, lay):
    n = C.c_int(0)
    lib.check(lib.lib().iamrx_layout_coalesced_boxes(lay.h, C.byref(n), None))
    arr = (C.c_int * (6 * n.value))()
    lib.check(lib.lib().iamrx_layout_coalesced_boxes(lay.h, C.byref(n), arr))
    return sorted((tuple(arr[6 * q:6 * q + 3]), tuple(arr[6 * q + 3:6 * q + 6])) for q in range(n.value))


def test_merging_rules(merged):
    lib = merged
    n = (32, 32, 32)
    assert coalesced(lib, lib.Layout.decompose(n, 8)) == [((0, 0, 0), (31, 31, 31))]                     # 64 boxes -> 1
    assert coalesced(lib, lib.Layout.decompose((48, 32, 16), (16, 32, 8))) == [((0, 0, 0), (47, 31, 15))]
    # L-shape: three boxes -> two (the pair sharing a full face), never a box that covers cells nobody owns
    L = lib.Layout([((0, 0, 0), (15, 15, 15)), ((16, 0, 0), (31, 15, 15)), ((0, 16, 0), (15, 31, 15))])
    got = coalesced(lib, L)
    assert len(got) == 2 and sum(np.prod([h - l + 1 for l, h in zip(lo, hi)]) for lo, hi in got) == 3 * 16 ** 3
    # boxes of different owners stay apart
    two = lib.Layout([((0, 0, 0), (15, 31, 31)), ((16, 0, 0), (31, 31, 31))], [0, 1])
    assert len(coalesced(lib, two)) == 2
    # faces that do not match: nothing merges
    odd = lib.Layout([((0, 0, 0), (15, 15, 15)), ((16, 0, 0), (31, 7, 15))])
    assert len(coalesced(lib, odd)) == 2
    lib.tuning_set("COALESCE", 0)
    assert len(coalesced(lib, lib.Layout.decompose(n, 8))) == 64


def run_tg(lib, n, mgs, nsteps, **kw):
    from iamr_amd import ns as NS
    g = lib.Geom.make(n)
    lay = lib.Layout.single(n) if mgs is None else lib.Layout.decompose(n, mgs)
    ns = NS.NavierStokes(g, lay, NS.ns_params(init_iter=2, cfl=0.7, visc_coef=1e-3, **kw))
    ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
    ns.post_init(-1.0)
    dts = [ns.step() for _ in range(nsteps)]
    return ns, lay, dts


def test_chopped_periodic_level_equals_the_single_box_run_to_the_bit(merged):
    lib = merged
    from iamr_amd import ns as NS
    n = (64, 64, 64)
    ns1, lay1, dt1 = run_tg(lib, n, None, 3)
    ns64, lay64, dt64 = run_tg(lib, n, 16, 3)
    assert dt1 == dt64 and len(lay64.boxes) == 64
    for sel, typ in ((NS.NavierStokes.S_NEW, None), (2, "node"), (4, None)):        # state, pressure, grad p
        A = ns1.data(sel)
        B = ns64.data(sel)
        assert B.nlocal() == 64                                                      # the accessor speaks the caller's boxes
        a = A.gather_valid(n)
        b = B.gather_valid(n)
        assert np.array_equal(a, b), (sel, float(np.abs(a - b).max()))
    # ghost cells of the caller's boxes are filled from the level (interior ones = the neighbour's valid data)
    S = ns64.data(NS.NavierStokes.S_NEW)
    full = ns1.data(NS.NavierStokes.S_NEW).gather_valid(n)
    arr, lo = S.to_numpy(21)
    blo, bhi, gi = lay64.local_box(21)
    idx = [np.mod(np.arange(blo[d] - 1, bhi[d] + 2), n[d]) for d in range(3)]
    assert np.array_equal(arr, full[np.ix_(*idx)])
    # derived quantity on the caller's layout
    e1, e64 = ns1.derive("energy").gather_valid(n), ns64.derive("energy").gather_valid(n)
    assert np.array_equal(e1, e64)
    # set_data on the caller's layout reaches the level: overwrite the state with the single-box run's and continue both
    ns64.set_data(NS.NavierStokes.S_NEW, S)
    assert ns1.step() == ns64.step()
    assert np.array_equal(ns1.data(0).gather_valid(n), ns64.data(0).gather_valid(n))


def test_wall_bounded_level_in_eight_boxes(merged):
    """LidDrivenCavity set-up (slip / no-slip walls, moving lid, tracer diffusion): 8 boxes merged = 1 box, bit for bit"""
    lib = merged
    from iamr_amd import ns as NS
    n = (32, 32, 32)
    res = []
    for mgs in (None, 16):
        g = lib.Geom.make(n, periodic=(0, 0, 0))
        lay = lib.Layout.single(n) if mgs is None else lib.Layout.decompose(n, mgs)
        ns = NS.NavierStokes(g, lay, NS.ns_params(phys_lo=[4, 4, 5], phys_hi=[5, 5, 5], wall_vel_hi=[0.0] * 6 + [1.0, 0.0, 0.0], cfl=0.3, visc_coef=0.01,
                                                  init_dt=0.0140625, init_shrink=0.3, init_iter=3, tracer_diff_coef=0.001))
        ns.init_rest(1.0)
        ns.post_init(-1.0)
        dts = [ns.step() for _ in range(3)]
        res.append((dts, ns.data(0).gather_valid(n), ns.data(2).gather_valid(n)))
    assert res[0][0] == res[1][0]
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])


def test_two_level_hierarchy_merges_only_the_level_that_covers_its_domain(merged):
    """viscous two-level run, coarse level in 8 boxes, refined level in 4: the coarse level (covers the domain) is merged, the refined level
    keeps the caller's boxes -- the reference's tensor operator fills edge / corner ghost cells at coarse/fine boundaries box by box, so
    its answer depends on the fine boxes at O(h^2) (2e-5 here) and merging them would answer another question.  Merged-coarse run ==
    all-boxes-kept run to solver tolerance; both report the caller's boxes."""
    lib = merged
    from iamr_amd import ns as NS
    from iamr_amd.amr import Amr
    n0 = 16
    g0 = lib.Geom.make((n0,) * 3)
    cb = [((i, j, k), (i + 7, j + 7, k + 7)) for k in (0, 8) for j in (0, 8) for i in (0, 8)]
    fb = [((8 + 8 * i, 8 + 8 * j, 8), (15 + 8 * i, 15 + 8 * j, 23)) for j in (0, 1) for i in (0, 1)]
    out = []
    for co in (1, 0):
        lib.tuning_set("COALESCE", co)
        lays = [lib.Layout(cb), lib.Layout(fb)]
        amr = Amr(g0, lays, NS.ns_params(cfl=0.7, visc_coef=0.01, tracer_diff_coef=0.005, init_iter=2), lib.mg_opts())
        for l in range(2):
            amr.levels[l].init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
        amr.post_init()
        dts = [amr.coarse_step() for _ in range(2)]
        assert [len(l.boxes) for l in amr.layouts] == [8, 4]
        fine = amr.levels[1].data(0)
        out.append((dts, amr.levels[0].data(0).gather_valid((n0,) * 3),
                    {lays[1].local_box(li)[2]: fine.to_numpy(li)[0][1:-1, 1:-1, 1:-1] for li in range(fine.nlocal())}))
    lib.tuning_set("COALESCE", 1)
    assert np.allclose(out[0][0], out[1][0], rtol=1e-10, atol=0)
    assert np.abs(out[0][1] - out[1][1]).max() <= 1e-8
    for k in out[0][2]:
        assert np.abs(out[0][2][k] - out[1][2][k]).max() <= 1e-8


def test_restart_of_a_chopped_run(merged, tmp_path, capsys):
    """checkpoints are written and read on the caller's boxes; the restarted run continues bit for bit"""
    from iamr_amd import run as R
    from iamr_amd.plotfile import PlotFile
    inp = os.path.join(HERE, "golden", "inputs.3d.taylorgreen")
    plt, chk = str(tmp_path / "plt"), str(tmp_path / "chk")
    args = [inp, "amr.n_cell=32 32 32", "max_step=6", "stop_time=100.0", "amr.plot_int=6", "amr.max_grid_size=8"]
    assert R.main(args + [f"amr.plot_file={plt}", f"amr.check_file={chk}", "amr.check_int=3"]) == 0
    plt2 = str(tmp_path / "rst")
    assert R.main(args + [f"amr.plot_file={plt2}", "amr.check_int=-1", f"amr.restart={chk}00003"]) == 0
    capsys.readouterr()
    A, B = PlotFile.read(plt + "00006"), PlotFile.read(plt2 + "00006")
    assert len(A.levels[0].boxes) == 64 and A.levels[0].boxes == B.levels[0].boxes
    for x, y in zip(A.levels[0].data, B.levels[0].data):
        assert np.array_equal(x, y)


def test_operator_entries_on_caller_owned_chopped_arrays(merged):
    """the L3 operator boundary as IAMR drives it (MacProj::mlmg_mac_solve, Projection::doMLMGNodalProjection on the level's MultiFabs, chopped at
    amr.max_grid_size): iamrx_mlmg_mac_solve and iamrx_nodal_projection solve on the merged boxes and hand the results back on the caller's --
    the answers of the same calls on a single box, to the bit"""
    lib = merged
    from iamr_amd import ns as N
    n = (64, 32, 32)
    g = lib.Geom.make(n, prob_hi=(2.0, 1.0, 1.0))
    rng = np.random.default_rng(2)
    ax = [(np.arange(-1, n[d] + 1) + 0.5) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    rho = 1.0 + 0.3 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) + 0.1 * np.cos(4 * np.pi * Z)
    res = []
    for mgs in (None, 16):
        lay = lib.Layout.single(n) if mgs is None else lib.Layout.decompose(n, mgs)
        um = []
        for d in range(3):
            t = lib.face(d)
            axf = [(np.arange(-1, n[e] + t[e] + 1) + (0.0 if t[e] else 0.5)) / n[e] for e in range(3)]
            Xf, Yf, Zf = np.meshgrid(*axf, indexing="ij")
            m = lib.MultiFab(lay, t, 1, 1)
            m.set_from_global((np.sin(2 * np.pi * Xf + d) * np.cos(2 * np.pi * Yf) + 0.5 * np.cos(2 * np.pi * Zf + 0.3 * d))[..., None], (-1,) * 3)
            um.append(m)
        rho_d = lib.MultiFab(lay, lib.CELL, 1, 1); rho_d.set_from_global(rho[..., None], (-1,) * 3)
        phi_d = lib.MultiFab(lay, lib.CELL, 1, 1); phi_d.setval(0.0)
        st = lib.mlmg_mac_solve(g, um, rho_d, 0, None, phi_d, 2.0 / 0.01, mac_tol=1e-11)
        assert st.converged
        div = lib.MultiFab(lay, lib.CELL, 1, 0)
        lib.mac_divergence(g, div, um)
        assert div.norm0() <= 1e-8
        # nodal projection of a cell-centred field with variable sigma
        vel = lib.MultiFab(lay, lib.CELL, 3, 1)
        V = np.stack([np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y), np.cos(2 * np.pi * X) * np.sin(4 * np.pi * Y) + 0.2 * np.sin(2 * np.pi * Z),
                      0.5 * np.sin(2 * np.pi * (X + Z))], axis=-1)
        vel.set_from_global(V, (-1,) * 3)
        sig = lib.MultiFab(lay, lib.CELL, 1, 1); sig.set_from_global((1.0 / rho)[..., None], (-1,) * 3)
        p = lib.MultiFab(lay, lib.NODE, 1, 1); p.setval(0.0)
        gp = lib.MultiFab(lay, lib.CELL, 3, 1); gp.setval(0.0)
        stn = N.nodal_projection(g, vel, 0, p, sig, 0, rel_tol=1e-11, gp=gp)
        assert stn.converged
        res.append((st.iters, stn.iters, phi_d.gather_valid(n), [m.gather_valid(n) for m in um], vel.gather_valid(n), p.gather_valid(n), gp.gather_valid(n)))
    a, b = res
    assert a[0] == b[0] and a[1] == b[1]
    assert np.array_equal(a[2], b[2])
    for x, y in zip(a[3], b[3]):
        assert np.array_equal(x, y)
    assert np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5]) and np.array_equal(a[6], b[6])
