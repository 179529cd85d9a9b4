"""Communicator set-up for the multi-GPU path (one process per GPU).

init_rccl_from_torch(dist): production transport -- RCCL inside libiamrx.so; torch.distributed is only used to
broadcast the 128-byte RCCL unique id.
init_gloo_callback(dist): test transport -- torch.distributed (gloo) send/recv/all_reduce on host buffers driven
from the library through C callbacks; lets two ranks share one GPU.
Call after lib.init() and before creating any Layout."""
import ctypes as C
import numpy as np
from .lib import lib, IamrxError

_keep = []


def _check(rc):
    if rc != 0:
        raise IamrxError(lib().iamrx_comm_last_error().decode() if hasattr(lib(), "iamrx_comm_last_error") else "comm error")


def init_rccl_from_torch(dist):
    import torch
    L = lib()
    L.iamrx_comm_last_error.restype = C.c_char_p
    rank, world = dist.get_rank(), dist.get_world_size()
    buf = (C.c_char * 128)()
    if rank == 0:
        _check(L.iamrx_comm_get_unique_id(buf))
    t = torch.tensor(list(bytes(buf)), dtype=torch.uint8, device="cuda")
    dist.broadcast(t, src=0)
    raw = bytes(t.cpu().tolist())
    idb = (C.c_char * 128).from_buffer_copy(raw)
    _check(L.iamrx_comm_init_rccl(idb, rank, world))


AR_CB = C.CFUNCTYPE(None, C.POINTER(C.c_double), C.c_int, C.c_int)
EX_CB = C.CFUNCTYPE(None, C.c_int, C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_long),
                    C.c_int, C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_double)), C.POINTER(C.c_long))


def init_gloo_callback(dist, group=None):
    import torch
    L = lib()
    L.iamrx_comm_last_error.restype = C.c_char_p
    rank, world = dist.get_rank(), dist.get_world_size()

    def allreduce(vals, n, op):
        a = np.ctypeslib.as_array(vals, shape=(n,))
        t = torch.from_numpy(a.copy())
        dist.all_reduce(t, op={0: dist.ReduceOp.SUM, 1: dist.ReduceOp.MAX, 2: dist.ReduceOp.MIN}[op], group=group)
        a[:] = t.numpy()

    def exchange(ns, speers, sbufs, scounts, nr, rpeers, rbufs, rcounts):
        reqs, recvs = [], []
        for i in range(nr):
            t = torch.empty(rcounts[i], dtype=torch.float64)
            recvs.append((t, rbufs[i], rcounts[i]))
            reqs.append(dist.irecv(t, rpeers[i], group=group, tag=0))
        for i in range(ns):
            a = np.ctypeslib.as_array(sbufs[i], shape=(scounts[i],))
            reqs.append(dist.isend(torch.from_numpy(a.copy()), speers[i], group=group, tag=0))
        for r in reqs:
            r.wait()
        for t, ptr, cnt in recvs:
            np.ctypeslib.as_array(ptr, shape=(cnt,))[:] = t.numpy()

    ar, ex = AR_CB(allreduce), EX_CB(exchange)
    _keep.extend([ar, ex])
    _check(L.iamrx_comm_init_callback(rank, world, ar, ex))
