"""GPU: the rank-aware (N > 1) code path of the product -- box ownership, packed halo messages, global
reductions inside the multigrid solvers and the time step -- executed by TWO ranks sharing the one GPU of the
test box (transport: CallbackComm over torch.distributed/gloo; the RCCL transport implements the same two
primitives).  The 2-rank result must equal the 1-rank result to solver tolerance."""
import os
import sys
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = (16, 16, 16)
BOXES = [((0, 0, 0), (15, 15, 7)), ((0, 0, 8), (15, 15, 15))]
NSTEPS = 2


def run(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
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
    owners = [0, 1] if world > 1 else [0, 0]
    lay = lib.Layout(BOXES, owners)
    g = lib.Geom.make(N)
    ns = NS.NavierStokes(g, lay, NS.ns_params(cfl=0.5, visc_coef=1e-2))
    ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
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


def test_two_ranks_on_one_gpu_match_single_rank(tmp_path):
    import torch.multiprocessing as mp
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(run, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    mp.spawn(run, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    ref = np.load(os.path.join(str(tmp_path), "w1_r0.npz"))
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), f"w2_r{r}.npz"))
        assert np.allclose(z["dts"], ref["dts"], rtol=1e-10, atol=0)
        assert np.array_equal(z["iters"], ref["iters"])
        key = f"box{r}"
        assert np.abs(z[key] - ref[key]).max() <= 1e-9, np.abs(z[key] - ref[key]).max()
