"""CPU suite: the N > 1 path on world_size-2 gloo.

The boxes of a level are split over two ranks; every ghost exchange is driven by the PRODUCT's host-side
planner (iamrx_host_fill_plan: local copies + packed peer messages) with torch.distributed(gloo) as the
transport, while the oracle supplies the per-box arithmetic (red-black Gauss-Seidel).  The distributed result
must be bit-identical to the single-box, single-rank sweep: this checks partitioning, message
matching/ordering and periodic images exactly as the RCCL path uses them."""
import ctypes as C
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = (16, 8, 8)
BOXES = [((i0, 0, 0), (i0 + 3, 7, 7)) for i0 in (0, 4, 8, 12)]
OWNERS = [0, 0, 1, 1]
NSWEEP = 3


def global_fields():
    rng = np.random.default_rng(42)
    phi = rng.standard_normal(N)
    rhs = rng.standard_normal(N)
    b = []
    for d in range(3):
        shp = list(N)
        shp[d] += 1
        f = 1.0 + 0.5 * rng.random(shp)
        hi = [slice(None)] * 3
        lo = [slice(None)] * 3
        hi[d] = N[d]
        lo[d] = 0
        f[tuple(hi)] = f[tuple(lo)]        # periodic duplicate face
        b.append(f)
    return phi, rhs, b


def reference_single_box(orc):
    L = orc.lib()
    g = orc.geom(N)
    phi0, rhs0, b0 = global_fields()
    b = []
    for d in range(3):
        f = orc.Fab(N, orc.face(d), 0, 1)
        f.a[..., 0] = b0[d]
        b.append(f)
    lev = orc.abec_level(g, b)
    phi = orc.Fab(N, orc.CELL, 1, 1)
    phi.valid(N)[..., 0] = phi0
    rhs = orc.Fab(N, orc.CELL, 0, 1)
    rhs.a[..., 0] = rhs0
    z = orc.i3([0, 0, 0])
    for _ in range(NSWEEP):
        for rb in (0, 1):
            L.orc_fill_periodic(phi.ref(), C.byref(g), orc.i3(orc.CELL))
            L.orc_abec_gsrb(C.byref(lev), phi.ref(), rhs.ref(), rb, C.c_double(1.15), z, z, 3)
    return phi.valid(N)[..., 0].copy()


def worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    import orc
    from iamr_amd import lib
    from test_cpu_abi import host_plan
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    L = orc.lib()
    geom = lib.Geom.make(N)
    desc = host_plan(lib, BOXES, OWNERS, rank, (0, 0, 0), 1, geom)
    phi0, rhs0, b0 = global_fields()
    mine = [i for i, o in enumerate(OWNERS) if o == rank]
    fabs, levs, keep = {}, {}, []
    for bi in mine:
        lo, hi = BOXES[bi]
        bn = tuple(hi[d] - lo[d] + 1 for d in range(3))
        bg = orc.geom(bn, periodic=(0, 0, 0))
        bg.dx = (C.c_double * 3)(*[1.0 / N[d] for d in range(3)])
        bf = []
        for d in range(3):
            f = orc.Fab(bn, orc.face(d), 0, 1)
            sl = tuple(slice(lo[q], hi[q] + 1 + (1 if q == d else 0)) for q in range(3))
            f.a[..., 0] = b0[d][sl]
            bf.append(f)
        phi = orc.Fab(bn, orc.CELL, 1, 1)
        sl = tuple(slice(lo[q], hi[q] + 1) for q in range(3))
        phi.valid(bn)[..., 0] = phi0[sl]
        rhs = orc.Fab(bn, orc.CELL, 0, 1)
        rhs.a[..., 0] = rhs0[sl]
        fabs[bi] = (phi, rhs, bn, lo)
        levs[bi] = (orc.abec_level(bg, bf), bg)
        keep.append(bf)

    def region_view(bi, lo_, hi_, shift=(0, 0, 0)):
        phi, _, bn, blo = fabs[bi]
        # fab array origin (global index) = blo - 1
        return tuple(slice(lo_[q] + shift[q] - (blo[q] - 1), hi_[q] + shift[q] - (blo[q] - 1) + 1) for q in range(3))

    def exchange():
        # pack + send / recv + unpack, then local copies (product order: execute_plan)
        peers = sorted(set(int(d[1]) for d in desc if d[0] != 0))
        reqs, recvbufs = [], {}
        for p in peers:
            pk = [d for d in desc if d[0] == 1 and d[1] == p]
            up = [d for d in desc if d[0] == 2 and d[1] == p]
            if pk:
                buf = np.concatenate([fabs[int(d[2])][0].a[region_view(int(d[2]), d[4:7], d[7:10], d[10:13])][..., 0].ravel(order="F") for d in pk])
                reqs.append(dist.isend(torch.from_numpy(buf.copy()), p))
            if up:
                npts = sum(int(np.prod(d[7:10] - d[4:7] + 1)) for d in up)
                t = torch.empty(npts, dtype=torch.float64)
                recvbufs[p] = (t, up)
                reqs.append(dist.irecv(t, p))
        for d in desc:
            if d[0] == 0:
                s, t = int(d[2]), int(d[3])
                fabs[t][0].a[region_view(t, d[4:7], d[7:10])] = fabs[s][0].a[region_view(s, d[4:7], d[7:10], d[10:13])]
        for r in reqs:
            r.wait()
        for p, (t, up) in recvbufs.items():
            a = t.numpy()
            for d in up:
                shp = tuple(int(x) for x in (d[7:10] - d[4:7] + 1))
                off = int(d[13])
                fabs[int(d[3])][0].a[region_view(int(d[3]), d[4:7], d[7:10])][..., 0] = a[off:off + int(np.prod(shp))].reshape(shp, order="F")

    z = orc.i3([0, 0, 0])
    for _ in range(NSWEEP):
        for rb in (0, 1):
            exchange()
            for bi in mine:
                phi, rhs, bn, lo = fabs[bi]
                lev, bg = levs[bi]
                # box origins are multiples of 4, so local and global colour parity coincide
                L.orc_abec_gsrb(C.byref(lev), phi.ref(), rhs.ref(), rb, C.c_double(1.15), z, z, 3)
    # a scalar all-reduce as used for norms / dt (ParallelDescriptor::ReduceRealMax role)
    loc = max(np.abs(fabs[bi][0].valid(fabs[bi][2])).max() for bi in mine)
    t = torch.tensor([loc], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), gmax=t.item(),
             **{f"box{bi}": fabs[bi][0].valid(fabs[bi][2])[..., 0] for bi in mine})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_halo_exchange_matches_single_rank(orc, tmp_path):
    import torch.multiprocessing as mp
    ref = reference_single_box(orc)
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:        # a free port (a pid-derived one can collide with a lingering socket)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mp.spawn(worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.zeros(N)
    gmax = []
    for r in range(2):
        z = np.load(os.path.join(str(tmp_path), f"rank{r}.npz"))
        gmax.append(float(z["gmax"]))
        for bi, (lo, hi) in enumerate(BOXES):
            if OWNERS[bi] == r:
                got[tuple(slice(lo[q], hi[q] + 1) for q in range(3))] = z[f"box{bi}"]
    assert np.array_equal(got, ref), np.abs(got - ref).max()
    assert gmax[0] == gmax[1] == np.abs(ref).max()
