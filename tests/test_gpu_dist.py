"""GPU: the rank-aware (N > 1) code path of the product -- box ownership, packed halo messages, global
reductions inside the multigrid solvers and the time step -- executed by TWO ranks sharing the one GPU of the
test box (transport: CallbackComm over torch.distributed/gloo; the RCCL transport implements the same two
primitives).  The 2-rank result must equal the 1-rank result to solver tolerance."""
import os
import sys
import numpy as np
import pytest

# one box per rank in most cases (nothing to merge); the cases with several boxes per rank say which mode they want ("+merge"), so the
# whole module runs with IAMRX_COALESCE = 0 in the environment of the rank processes unless a case sets it (tests/conftest.py)
pytestmark = [pytest.mark.gpu, pytest.mark.boxes_kept]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = (16, 16, 16)
BOXES = [((0, 0, 0), (15, 15, 7)), ((0, 0, 8), (15, 15, 15))]
NSTEPS = 2


def free_port():
    """a TCP port nobody listens on right now (a port derived from the pid collided with a lingering socket once in round 6: the rendezvous
    of one test then waited ten minutes inside the driver's time budget)"""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run(rank, world, port, out_dir, agg, case="tg"):
    sys.path.insert(0, ROOT)
    global N, BOXES
    if case.endswith("+merge"):      # the library's default: the level objects work on the boxes a rank owns MERGED (the suite otherwise keeps them)
        os.environ["IAMRX_COALESCE"] = "1"
        case = case[:-6]
    if case == "slabs":              # eight 16^3 boxes, the four of a z-slab owned by one rank: each rank merges its four into one 32 x 32 x 16 box
        N = (32, 32, 32)
        BOXES = [((16 * i, 16 * j, 16 * k), (16 * i + 15, 16 * j + 15, 16 * k + 15)) for k in (0, 1) for j in (0, 1) for i in (0, 1)]
    if case == "stack4":     # the weak-scaling layout of bench.py: one box per rank, stacked in z
        N = (16, 16, 64)
        BOXES = [((0, 0, 16 * r), (15, 15, 16 * r + 15)) for r in range(4)]
    if case == "rows128":    # boxes whose rows span the domain, split in y and z over the ranks: the multi-box red + black sweep with its two-layer
        N = (128, 96, 64)    # exchange, issued in two parts around the exchange (interior tiles on the main stream, the rest on the side stream)
        BOXES = [((0, 48 * (q % 2), 32 * (q // 2)), (127, 48 * (q % 2) + 47, 32 * (q // 2) + 31)) for q in range(4)]
    if case.startswith("grid"):   # bench.py's layout for N ranks: one 16^3 box per rank on the most cubic process grid (4 -> 2x2x1, 8 -> 2x2x2)
        import bench
        nr = int(case[4:])
        pg = bench.proc_grid(nr)
        N = tuple(16 * pg[d] for d in range(3))
        BOXES = []
        for r in range(nr):
            ix, iy, iz = r % pg[0], (r // pg[0]) % pg[1], r // (pg[0] * pg[1])
            BOXES.append(((16 * ix, 16 * iy, 16 * iz), (16 * ix + 15, 16 * iy + 15, 16 * iz + 15)))
    if agg is not None:
        os.environ["IAMRX_MG_AGGLOMERATE_CELLS"] = agg
    from iamr_amd import lib
    from iamr_amd import ns as NS
    lib.init(0)
    if world > 1:
        import torch.distributed as dist
        from iamr_amd import comm
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        comm.init_gloo_callback(dist)
        import bench
        bench.transport_selftest(lib, rank, world)   # the check bench.py runs on a freshly initialised transport
    if case == "amr":
        return run_amr(rank, world, out_dir)
    if case.startswith("amrbench"):
        return run_amrbench(rank, world, out_dir, int(case[8:]))
    if case == "regrid":
        return run_regrid(rank, world, out_dir)
    owners = list(range(len(BOXES))) if world > 1 else [0] * len(BOXES)     # case 'tg' on 3 ranks: rank 2 owns no box
    if case == "rows128":
        owners = [q % world for q in range(len(BOXES))]
    if case == "slabs":
        owners = [(q // 4) % world for q in range(len(BOXES))]
    lay = lib.Layout(BOXES, owners)
    if case.startswith("grid"):
        g = lib.Geom.make(N, prob_hi=tuple(N[d] / 16.0 for d in range(3)))
        ns = NS.NavierStokes(g, lay, NS.ns_params(cfl=0.7, visc_coef=1e-3, init_iter=2))
        ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
    elif case == "rows128":
        g = lib.Geom.make(N, prob_hi=(1.0, 0.75, 0.5))
        ns = NS.NavierStokes(g, lay, NS.ns_params(cfl=0.7, visc_coef=1e-3, init_iter=2))
        ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
    elif case == "stack4":
        g = lib.Geom.make(N, prob_hi=(1.0, 1.0, 4.0))
        ns = NS.NavierStokes(g, lay, NS.ns_params(cfl=0.7, visc_coef=1e-3, init_iter=2))
        ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
    elif case in ("tg", "slabs"):
        g = lib.Geom.make(N)
        ns = NS.NavierStokes(g, lay, NS.ns_params(cfl=0.5, visc_coef=1e-2))
        ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
    else:   # lid-driven cavity: walls in every direction, the lid sits on rank 1's box
        g = lib.Geom.make(N, periodic=(0, 0, 0))
        lid = [0.0] * 9
        lid[6] = 1.0
        ns = NS.NavierStokes(g, lay, NS.ns_params(cfl=0.3, visc_coef=0.01, init_dt=0.0140625, init_shrink=0.3, init_iter=3, tracer_diff_coef=0.001,
                                                  phys_lo=[4, 4, 5], phys_hi=[5, 5, 5], wall_vel_hi=lid))
        ns.init_rest(1.0)
    ns.post_init(-1.0)
    dts = [ns.step() for _ in range(NSTEPS)]
    S = ns.data(NS.NavierStokes.S_NEW)
    out = {"dts": np.array(dts)}
    for li in range(S.nlocal()):
        a, lo = S.to_numpy(li)
        blo, bhi, gi = lay.local_box(li)
        out[f"box{gi}"] = a[1:-1, 1:-1, 1:-1, :]
    sm, sn, sv = ns.stats()
    out["iters"] = np.array([sm.iters, sn.iters, sv.iters])
    np.savez(os.path.join(out_dir, f"w{world}_r{rank}.npz"), **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_amr(rank, world, out_dir):
    """two-level hierarchy (viscous TaylorGreen): coarse level in 8 boxes, refined level in 4 boxes, both spread over the ranks"""
    from iamr_amd import lib
    from iamr_amd import ns as NS
    from iamr_amd.amr import Amr
    import torch.distributed as dist
    n0 = 16
    cb = [((i, j, k), (i + 7, j + 7, k + 7)) for k in (0, 8) for j in (0, 8) for i in (0, 8)]
    fb = [((8 + 8 * i, 8 + 8 * j, 8), (15 + 8 * i, 15 + 8 * j, 23)) for j in (0, 1) for i in (0, 1)]
    own = (lambda n: [q % world for q in range(n)]) if world > 1 else (lambda n: [0] * n)
    lays = [lib.Layout(cb, own(len(cb))), lib.Layout(fb, own(len(fb)))]
    amr = Amr(lib.Geom.make((n0,) * 3), lays, NS.ns_params(cfl=0.7, visc_coef=0.01, tracer_diff_coef=0.005, init_iter=2), lib.mg_opts())
    for l in range(2):
        amr.levels[l].init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
    amr.post_init()
    dts = [amr.coarse_step() for _ in range(2)]
    out = {"dts": np.array(dts)}
    for l in range(2):
        S = amr.levels[l].data(0)
        for li in range(S.nlocal()):
            a, lo = S.to_numpy(li)
            blo, bhi, gi = lays[l].local_box(li)
            out[f"l{l}box{gi}"] = a[1:-1, 1:-1, 1:-1, :]
    st, stm = amr.sync_stats()
    out["iters"] = np.array([st.iters, stm.iters])
    np.savez(os.path.join(out_dir, f"amr_w{world}_r{rank}.npz"), **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_amrbench(rank, world, out_dir, nbox):
    """bench.py's 2-level AMR workload in the layout it builds for `nbox` GPUs (one base box + one refined box per GPU on the process
    grid), spread over the `world` ranks present"""
    from iamr_amd import lib
    import bench
    import torch.distributed as dist
    keep = []
    res = bench.amr_workload(lib, 16, 1, rank, world, dist if world > 1 else None, layout_gpus=nbox, keep=keep)
    amr = keep[0]
    out = {"rate": np.array([res["cells_advanced_per_sec"]]), "iters": np.array([res["sync_project_iters"], res["mac_sync_iters"]]), "dts": np.array(amr.dts())}
    for l in range(2):
        S = amr.levels[l].data(0)
        for li in range(S.nlocal()):
            a, lo = S.to_numpy(li)
            blo, bhi, gi = amr.layouts[l].local_box(li)
            out[f"l{l}box{gi}"] = a[1:-1, 1:-1, 1:-1, :]
    np.savez(os.path.join(out_dir, f"amrbench_w{world}_r{rank}.npz"), **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def run_regrid(rank, world, out_dir):
    """a tracer blob drifting through the Taylor-Green flow, level 1 = where tracer > 0.3, regridded at the start of every coarse step:
    tag maps combined over the ranks, new boxes dealt out by the knapsack, data of the new level moved between ranks"""
    from iamr_amd import lib
    from iamr_amd import ns as NS
    from iamr_amd.amr import Amr
    import torch.distributed as dist
    n0 = 16
    cb = [((i, j, k), (i + 7, j + 7, k + 7)) for k in (0, 8) for j in (0, 8) for i in (0, 8)]
    fb = [((8, 8, 8), (23, 23, 23))]
    own = (lambda n: [q % world for q in range(n)])
    lays = [lib.Layout(cb, own(len(cb))), lib.Layout(fb, own(len(fb)))]
    amr = Amr(lib.Geom.make((n0,) * 3), lays, NS.ns_params(cfl=0.7, visc_coef=0.0, init_iter=2), lib.mg_opts())

    def state(X, Y, Z):
        S = np.zeros(X.shape + (5,), order="F")
        S[..., 0] = 1.0 + np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) * np.cos(2 * np.pi * Z)
        S[..., 1] = -np.cos(2 * np.pi * X) * np.sin(2 * np.pi * Y) * np.cos(2 * np.pi * Z)
        S[..., 3] = 1.0
        S[..., 4] = np.exp(-((X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.5) ** 2) / 0.01)
        return S
    for l in range(2):
        lev = amr.levels[l]
        n = n0 * 2 ** l
        m = lib.MultiFab(lev.layout, lib.CELL, 5, 1)
        for li in range(m.nlocal()):
            lo, hi = m.fab_box(li)
            c = [(np.arange(lo[d], hi[d] + 1) + 0.5) / n for d in range(3)]
            m.from_numpy(state(*np.meshgrid(*c, indexing="ij")), li)
        lev.set_data(lev.S_NEW, m)
    amr.set_regrid(max_level=1, regrid_int=1, rules=[dict(comp=4, mode=0, value=[0.3])], blocking_factor=4, max_grid_size=8, n_error_buf=1)
    amr.post_init()
    out = {"dts": []}
    grids = []
    for step in range(3):
        out["dts"].append(amr.coarse_step())
        grids.append(sorted(amr.layouts[1].boxes))
    out["dts"] = np.array(out["dts"])
    out["grids"] = np.array([v for g in grids for lo, hi in g for v in (*lo, *hi)] + [len(g) for g in grids])
    for l in range(amr.nlev):
        S = amr.levels[l].data(0)
        for li in range(S.nlocal()):
            a, lo = S.to_numpy(li)
            blo, bhi, gi = amr.layouts[l].local_box(li)
            out["l%dbox_%d_%d_%d" % ((l,) + tuple(blo))] = a[1:-1, 1:-1, 1:-1, :]
    out["nlocal"] = np.array([amr.levels[l].data(0).nlocal() for l in range(amr.nlev)])
    np.savez(os.path.join(out_dir, f"regrid_w{world}_r{rank}.npz"), **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("nr", [2, 4])
def test_amr_bench_workload_sharded_over_ranks(tmp_path, nr):
    """the AMR workload of bench.py --gpus nr (base level and refined level each one box per GPU) on nr ranks sharing the GPU: every box of
    both levels, the time steps and the iteration counts of the sync solves equal the run of ONE rank holding the same 2 nr boxes"""
    import torch.multiprocessing as mp
    port = free_port()
    mp.spawn(run, args=(1, port, str(tmp_path), None, f"amrbench{nr}"), nprocs=1, join=True)
    ref = np.load(os.path.join(str(tmp_path), "amrbench_w1_r0.npz"))
    mp.spawn(run, args=(nr, free_port(), str(tmp_path), None, f"amrbench{nr}"), nprocs=nr, join=True)
    seen = set()
    for r in range(nr):
        z = np.load(os.path.join(str(tmp_path), f"amrbench_w{nr}_r{r}.npz"))
        assert np.allclose(z["dts"], ref["dts"], rtol=1e-12, atol=0)
        assert np.array_equal(z["iters"], ref["iters"])
        for key in z.files:
            if "box" in key:
                seen.add(key)
                assert np.abs(z[key] - ref[key]).max() <= 1e-9, (key, np.abs(z[key] - ref[key]).max())
    assert seen == {k for k in ref.files if "box" in k}


@pytest.mark.parametrize("merge", ["", "+merge"])
def test_regrid_on_two_ranks_matches_one_rank(tmp_path, merge):
    """Amr::regrid with the levels spread over two ranks: same grids after every regrid, same data, both ranks own boxes of the new level"""
    import torch.multiprocessing as mp
    port = free_port()
    mp.spawn(run, args=(1, port, str(tmp_path), None, "regrid" + merge), nprocs=1, join=True)      # "+merge": level 0 merged per rank (the default mode)
    ref = np.load(os.path.join(str(tmp_path), "regrid_w1_r0.npz"))
    mp.spawn(run, args=(2, free_port(), str(tmp_path), None, "regrid" + merge), nprocs=2, join=True)
    seen = set()
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), f"regrid_w2_r{r}.npz"))
        assert np.array_equal(z["grids"], ref["grids"])
        assert np.allclose(z["dts"], ref["dts"], rtol=1e-10, atol=0)
        assert z["nlocal"][1] >= 1
        for key in z.files:
            if "box" in key:
                seen.add(key)
                assert np.abs(z[key] - ref[key]).max() <= 1e-8, (key, np.abs(z[key] - ref[key]).max())
    assert seen == {k for k in ref.files if "box" in k}


def test_two_level_hierarchy_on_two_ranks(tmp_path):
    """the multi-level time step (subcycling, registers, MAC sync incl. the viscous solves, composite sync projection, multi-level
    initialisation) with the boxes of BOTH levels distributed over two ranks sharing the GPU: every box equals the one-rank result"""
    import torch.multiprocessing as mp
    port = free_port()
    mp.spawn(run, args=(1, port, str(tmp_path), None, "amr"), nprocs=1, join=True)
    ref = np.load(os.path.join(str(tmp_path), "amr_w1_r0.npz"))
    mp.spawn(run, args=(2, free_port(), str(tmp_path), None, "amr"), nprocs=2, join=True)
    seen = set()
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), f"amr_w2_r{r}.npz"))
        assert np.allclose(z["dts"], ref["dts"], rtol=1e-10, atol=0)
        for key in z.files:
            if "box" in key:
                seen.add(key)
                assert np.abs(z[key] - ref[key]).max() <= 1e-8, (key, np.abs(z[key] - ref[key]).max())
    assert seen == {k for k in ref.files if "box" in k}


@pytest.mark.parametrize("agg", [None, "0", "64"])
def test_two_ranks_on_one_gpu_match_single_rank(tmp_path, agg):
    """agg: multigrid agglomeration threshold in cells (None: default, every coarse level of this small problem is replicated on
    both ranks; "0": all levels stay distributed; "64": only the 4^3 and 2^3 levels are replicated)"""
    import torch.multiprocessing as mp
    port = free_port()
    mp.spawn(run, args=(1, port, str(tmp_path), agg), nprocs=1, join=True)
    mp.spawn(run, args=(2, free_port(), str(tmp_path), agg), nprocs=2, join=True)
    ref = np.load(os.path.join(str(tmp_path), "w1_r0.npz"))
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), f"w2_r{r}.npz"))
        assert np.allclose(z["dts"], ref["dts"], rtol=1e-10, atol=0)
        assert np.array_equal(z["iters"], ref["iters"])
        key = f"box{r}"
        assert np.abs(z[key] - ref[key]).max() <= 1e-9, np.abs(z[key] - ref[key]).max()


def test_a_rank_without_boxes_takes_part_in_the_step(tmp_path):
    """more ranks than boxes (AMReX allows it; coarse AMR levels routinely have fewer grids than ranks): rank 2 owns nothing, launches
    no kernels, but joins every reduction and exchange; ranks 0 and 1 reproduce the 1-rank result"""
    import torch.multiprocessing as mp
    port = free_port()
    mp.spawn(run, args=(1, port, str(tmp_path), None), nprocs=1, join=True)
    mp.spawn(run, args=(3, free_port(), str(tmp_path), None), nprocs=3, join=True)
    ref = np.load(os.path.join(str(tmp_path), "w1_r0.npz"))
    idle = np.load(os.path.join(str(tmp_path), "w3_r2.npz"))
    assert not [k for k in idle.files if k.startswith("box")] and np.allclose(idle["dts"], ref["dts"], rtol=1e-10, atol=0)
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), f"w3_r{r}.npz"))
        assert np.allclose(z["dts"], ref["dts"], rtol=1e-10, atol=0)
        assert np.array_equal(z["iters"], ref["iters"])
        key = f"box{r}"
        assert np.abs(z[key] - ref[key]).max() <= 1e-9, np.abs(z[key] - ref[key]).max()


def test_ranks_that_merge_their_boxes(tmp_path):
    """the library's default mode (IAMRX_COALESCE = 1) under several ranks: every rank owns four boxes that merge into
    one slab; the merged level objects, their data accessors (the caller's boxes) and the multigrid wrappers on merged boxes against the
    one-rank run, whose eight boxes merge into one box spanning the domain"""
    import torch.multiprocessing as mp
    port = free_port()
    mp.spawn(run, args=(1, port, str(tmp_path), None, "slabs+merge"), nprocs=1, join=True)
    mp.spawn(run, args=(2, free_port(), str(tmp_path), None, "slabs+merge"), nprocs=2, join=True)
    ref = np.load(os.path.join(str(tmp_path), "w1_r0.npz"))
    seen = 0
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), f"w2_r{r}.npz"))
        assert np.allclose(z["dts"], ref["dts"], rtol=1e-10, atol=0)
        for key in [k for k in z.files if k.startswith("box")]:
            assert np.abs(z[key] - ref[key]).max() <= 1e-9, (key, float(np.abs(z[key] - ref[key]).max()))
            seen += 1
    assert seen == 8          # the accessors speak the caller's eight boxes


@pytest.mark.parametrize("nr", [2, 4])
def test_rows_that_span_the_domain_sharded_in_y_and_z(tmp_path, nr):
    """four boxes of 128 x 48 x 32 (the shape of bench.py's multi-GPU layout: rows never cut) over 2 and 4 ranks: the MAC and viscous solves
    run the multi-box red + black sweep with ONE two-layer ghost exchange per sweep, issued on the side stream while the tiles that read
    no ghost cell run on the main stream (HALO_OVERLAP: on by itself where a level has peers); the coarse multigrid levels are agglomerated
    onto one merged box per rank.  Equal to the one-rank run (same kernels on four local boxes) to round-off: the sums of the mean
    removals and dot products are formed rank by rank."""
    import torch.multiprocessing as mp
    port = free_port()
    mp.spawn(run, args=(1, port, str(tmp_path), None, "rows128"), nprocs=1, join=True)
    mp.spawn(run, args=(nr, free_port(), str(tmp_path), None, "rows128"), nprocs=nr, join=True)
    ref = np.load(os.path.join(str(tmp_path), "w1_r0.npz"))
    for r in range(nr):
        z = np.load(os.path.join(str(tmp_path), f"w{nr}_r{r}.npz"))
        assert np.allclose(z["dts"], ref["dts"], rtol=1e-12, atol=0)
        assert np.array_equal(z["iters"], ref["iters"])
        for key in [k for k in z.files if k.startswith("box")]:
            assert np.abs(z[key] - ref[key]).max() <= 1e-12, (key, float(np.abs(z[key] - ref[key]).max()))


def test_two_ranks_lid_driven_cavity(tmp_path):
    """wall BCs (physical BC fills, Neumann projections, per-component tensor BCs, tracer diffusion) across a rank boundary"""
    import torch.multiprocessing as mp
    port = free_port()
    mp.spawn(run, args=(1, port, str(tmp_path), None, "ldc"), nprocs=1, join=True)
    mp.spawn(run, args=(2, free_port(), str(tmp_path), None, "ldc"), nprocs=2, join=True)
    ref = np.load(os.path.join(str(tmp_path), "w1_r0.npz"))
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), f"w2_r{r}.npz"))
        assert np.allclose(z["dts"], ref["dts"], rtol=1e-10, atol=0)
        assert np.array_equal(z["iters"], ref["iters"])
        key = f"box{r}"
        assert np.abs(z[key] - ref[key]).max() <= 1e-9, np.abs(z[key] - ref[key]).max()


def test_four_ranks_stacked_boxes_like_the_bench(tmp_path):
    """bench.py's weak-scaling layout (one box per rank stacked in z, domain 16 x 16 x 64) on 4 ranks sharing the GPU, with the
    default agglomeration threshold and with a threshold that keeps the first coarse level distributed: same result as one rank"""
    import torch.multiprocessing as mp
    port = free_port()
    mp.spawn(run, args=(1, port, str(tmp_path), None, "stack4"), nprocs=1, join=True)
    ref = np.load(os.path.join(str(tmp_path), "w1_r0.npz"))
    for agg in (None, "1024"):
        mp.spawn(run, args=(4, free_port(), str(tmp_path), agg, "stack4"), nprocs=4, join=True)
        for r in range(4):
            z = np.load(os.path.join(str(tmp_path), f"w4_r{r}.npz"))
            assert np.allclose(z["dts"], ref["dts"], rtol=1e-10, atol=0)
            assert np.array_equal(z["iters"], ref["iters"])
            key = f"box{r}"
            assert np.abs(z[key] - ref[key]).max() <= 1e-9, np.abs(z[key] - ref[key]).max()


@pytest.mark.parametrize("nr", [4, 8])
def test_bench_process_grid_layouts(tmp_path, nr):
    """the box layout bench.py builds for 4 / 8 GPUs (proc_grid: 2x2x1 / 2x2x2, every rank has a neighbour in each refined direction,
    SURVEY 8e) on nr ranks sharing the GPU over the callback transport: same result as one rank holding all boxes"""
    import torch.multiprocessing as mp
    port = free_port()
    mp.spawn(run, args=(1, port, str(tmp_path), None, f"grid{nr}"), nprocs=1, join=True)
    ref = np.load(os.path.join(str(tmp_path), "w1_r0.npz"))
    mp.spawn(run, args=(nr, free_port(), str(tmp_path), None, f"grid{nr}"), nprocs=nr, join=True)
    for r in range(nr):
        z = np.load(os.path.join(str(tmp_path), f"w{nr}_r{r}.npz"))
        assert np.allclose(z["dts"], ref["dts"], rtol=1e-10, atol=0)
        key = f"box{r}"
        assert np.abs(z[key] - ref[key]).max() <= 1e-9, np.abs(z[key] - ref[key]).max()


def test_bench_script_multi_rank_path(tmp_path):
    """bench.py launched the way the driver launches it for N > 1 (torch.distributed.run, one process per rank), here with 2 ranks
    sharing the GPU over the host-staged transport (IAMRX_BENCH_TRANSPORT=gloo): the script must run to the end on every rank and
    rank 0 must print one JSON line with the whole-job aggregate"""
    import json
    import subprocess
    env = dict(os.environ, IAMRX_BENCH_TRANSPORT="gloo", MASTER_ADDR="127.0.0.1", IAMRX_BENCH_N="32")     # "--n" would be eaten by torchrun
    port = free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["cells"] == 2 * 32 ** 3 and "cpu_baseline" not in d and d["transport"].startswith("gloo")
    assert abs(d["value"] - d["config"]["cells"] * 2 / (d["ms_per_step"] * 2e-3)) < 1e-6 * d["value"]


def _rccl_single(rank, out_dir):
    sys.path.insert(0, ROOT)
    import ctypes as C
    from iamr_amd import lib
    from iamr_amd import ns as NS
    lib.init(0)
    Lb = lib.lib()
    Lb.iamrx_comm_last_error.restype = C.c_char_p
    buf = (C.c_char * 128)()
    assert Lb.iamrx_comm_get_unique_id(buf) == 0, Lb.iamrx_comm_last_error()
    assert Lb.iamrx_comm_init_rccl(buf, 0, 1) == 0, Lb.iamrx_comm_last_error()
    # ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd, looped back to the own rank
    assert Lb.iamrx_comm_probe_exchange(0, C.c_long(100000)) == 0, Lb.iamrx_comm_last_error()
    n = (16, 16, 16)
    g = lib.Geom.make(n)
    lay = lib.Layout.decompose(n, 8)
    ns = NS.NavierStokes(g, lay, NS.ns_params(cfl=0.5, visc_coef=1e-2))
    ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
    ns.post_init(-1.0)
    dts = [ns.step() for _ in range(2)]
    S = ns.data(NS.NavierStokes.S_NEW).gather_valid(n)
    np.savez(os.path.join(out_dir, "rccl1.npz"), dts=np.array(dts), S=S)


def test_rccl_transport_single_rank(tmp_path):
    """the RCCL back end itself (dlopen of librccl, ncclGetUniqueId / ncclCommInitRank / ncclAllReduce through the resolved
    symbols) with a 1-rank communicator: every reduction of two time steps goes through ncclAllReduce; same result as the serial
    communicator; the halo-exchange primitive (grouped ncclSend / ncclRecv) is exercised as a loop-back to the own rank."""
    import torch.multiprocessing as mp
    mp.spawn(_rccl_single, args=(str(tmp_path),), nprocs=1, join=True)
    z = np.load(os.path.join(str(tmp_path), "rccl1.npz"))
    sys.path.insert(0, ROOT)
    from iamr_amd import lib
    from iamr_amd import ns as NS
    lib.init(0)
    n = (16, 16, 16)
    ns = NS.NavierStokes(lib.Geom.make(n), lib.Layout.decompose(n, 8), NS.ns_params(cfl=0.5, visc_coef=1e-2))
    ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
    ns.post_init(-1.0)
    dts = [ns.step() for _ in range(2)]
    assert np.array_equal(np.array(dts), z["dts"])
    assert np.array_equal(ns.data(NS.NavierStokes.S_NEW).gather_valid(n), z["S"])


def _rccl_pair(rank, out_dir):
    """one process per DEVICE (rank r drives device r): the RCCL transport between two GPUs -- ncclCommInitRank from a unique id handed over
    through a file, the exchange probe, bench.py's transport self-test, then two TaylorGreen steps on one 16 x 16 x 8 box per rank"""
    sys.path.insert(0, ROOT)
    import time
    from iamr_amd import lib
    from iamr_amd import ns as NS
    lib.init(rank)
    Lb = lib.lib()
    Lb.iamrx_comm_last_error.restype = C.c_char_p
    idf = os.path.join(out_dir, "rccl_id.bin")
    buf = (C.c_char * 128)()
    if rank == 0:
        assert Lb.iamrx_comm_get_unique_id(buf) == 0, Lb.iamrx_comm_last_error()
        with open(idf + ".tmp", "wb") as f:
            f.write(bytes(buf))
        os.replace(idf + ".tmp", idf)
    else:
        t0 = time.time()
        while not os.path.exists(idf):
            assert time.time() - t0 < 120, "rank 0 never published the RCCL unique id"
            time.sleep(0.05)
        buf = (C.c_char * 128).from_buffer_copy(open(idf, "rb").read())
    assert Lb.iamrx_comm_init_rccl(buf, rank, 2) == 0, Lb.iamrx_comm_last_error()
    assert Lb.iamrx_comm_probe_exchange(1 - rank, C.c_long(1 << 20)) == 0, Lb.iamrx_comm_last_error()
    import bench
    bench.transport_selftest(lib, rank, 2)
    lay = lib.Layout(BOXES, [0, 1])
    ns = NS.NavierStokes(lib.Geom.make(N), lay, NS.ns_params(cfl=0.5, visc_coef=1e-2))
    ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
    ns.post_init(-1.0)
    dts = [ns.step() for _ in range(NSTEPS)]
    S = ns.data(NS.NavierStokes.S_NEW)
    a, lo = S.to_numpy(0)
    np.savez(os.path.join(out_dir, f"rccl2_r{rank}.npz"), dts=np.array(dts), box=a[1:-1, 1:-1, 1:-1, :])


def test_rccl_transport_between_two_devices(tmp_path):
    """the first thing a multi-GPU lease should run.  Needs two visible devices; on the one-GPU development /
    grading box it is SKIPPED LOUDLY (the skip reason and a line on stderr say that RCCL has still only run as a one-rank loop-back)."""
    import torch
    nd = torch.cuda.device_count()
    if nd < 2:
        msg = (f"RCCL BETWEEN TWO DEVICES NOT EXERCISED: {nd} device(s) visible -- grouped ncclSend/ncclRecv + ncclAllReduce across xGMI "
               "remain untested on hardware (tests/test_gpu_dist.py::test_rccl_transport_between_two_devices runs as soon as 2 GPUs are visible)")
        print("\n*** " + msg + " ***", file=sys.stderr, flush=True)
        pytest.skip(msg)
    import torch.multiprocessing as mp
    mp.spawn(_rccl_pair, args=(str(tmp_path),), nprocs=2, join=True)
    mp.spawn(run, args=(1, free_port(), str(tmp_path), None), nprocs=1, join=True)      # the same two boxes on one rank
    one = np.load(os.path.join(str(tmp_path), "w1_r0.npz"))
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), f"rccl2_r{r}.npz"))
        assert np.allclose(z["dts"], one["dts"], rtol=1e-12, atol=0.0)
        ref = one[f"box{r}"]
        assert np.abs(z["box"] - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
