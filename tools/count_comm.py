"""counts halo exchanges / all-reduces per time step of the bench workload on 2 ranks sharing one GPU (scratch tool):
   python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29777 tools/count_comm.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch.distributed as dist
from iamr_amd import lib, comm
from iamr_amd import ns as N
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
single = os.environ.get('COUNT_SINGLE_OWNER') == '1'      # same boxes, all owned by rank 0 (run with one process)
lib.init(0)
cnt = {"ex": 0, "ar": 0, "bytes": 0}
import collections
hist = collections.Counter()
orig_irecv, orig_ar = dist.irecv, dist.all_reduce
def irecv(t, *a, **k):
    cnt["ex"] += 1; cnt["bytes"] += t.numel() * 8; hist[t.numel()] += 1
    return orig_irecv(t, *a, **k)
def ar(t, *a, **k):
    cnt["ar"] += 1
    return orig_ar(t, *a, **k)
dist.irecv, dist.all_reduce = irecv, ar
comm.init_gloo_callback(dist)
boxes = [((0, 0, r * n), (n - 1, n - 1, (r + 1) * n - 1)) for r in range(world)]
nb = int(os.environ.get('COUNT_BOXES', str(world)))
boxes = [((0, 0, r * n), (n - 1, n - 1, (r + 1) * n - 1)) for r in range(nb)]
lay = lib.Layout(boxes, [r % world for r in range(nb)])
g = lib.Geom.make((n, n, n * nb), prob_hi=(1.0, 1.0, float(nb)))
ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.7, visc_coef=1e-4, init_iter=2, init_shrink=1.0), lib.mg_opts())
ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
ns.post_init(-1.0)
ns.step()
for k in cnt: cnt[k] = 0
hist.clear()
import ctypes as C
def nsync():
    v = C.c_size_t(); lib.check(lib.lib().iamrx_sync_count(C.byref(v))); return v.value
def xc():
    v = (C.c_size_t * 4)(); lib.check(lib.lib().iamrx_exchange_counts(v)); return list(v)
s0 = nsync(); x0 = xc()
ns.step()
syncs = nsync() - s0
x1 = xc(); xd = [b - a for a, b in zip(x0, x1)]
sm, sn, sv = ns.stats()
if rank == 0:
    print(f"n={n} world={world}: per step and rank: {cnt['ex']} peer messages received ({cnt['bytes']/1e6:.1f} MB), {cnt['ar']} all-reduces; "
          f"{syncs} host synchronisations (the host-staged test transport adds two per all-reduce; RCCL reduces in place on the stream); "
          f"MG iterations mac {sm.iters} nodal {sn.iters} visc {sv.iters}")
if rank == 0:
    print(f"exchanges per step and rank: {xd[0]} on the main stream ({xd[1] * 8 / 1e6:.1f} MB sent; exposed: in front of the kernel that reads the ghost data), "
          f"{xd[2]} on the side stream ({xd[3] * 8 / 1e6:.1f} MB sent; hidden behind the interior tiles of the multi-box red + black sweep and of the nodal passes)")
if rank == 0: print("sizes (doubles: count):", sorted(hist.items()))
dist.barrier(); dist.destroy_process_group()
