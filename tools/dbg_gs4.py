import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from iamr_amd import lib
from iamr_amd import ns as N
lib.init(0)
n = (16, 16, 16)
g = lib.Geom.make(n); lay = lib.Layout.single(n)
rng = np.random.default_rng(5)
Xg = rng.standard_normal(n); Rg = rng.standard_normal(n); Sg = 1.0 / (1.0 + 0.8 * rng.random(n))
def node_field(G, ng):
    idx = [np.mod(np.arange(-ng, n[d] + 1 + ng), n[d]) for d in range(3)]
    return G[np.ix_(*idx)][..., None]
def cell_field(G, ng):
    idx = [np.mod(np.arange(-ng, n[d] + ng), n[d]) for d in range(3)]
    return G[np.ix_(*idx)][..., None]
res = []
for fused in (0, 1):
    ng = 4
    x = lib.MultiFab(lay, lib.NODE, 1, ng); r = lib.MultiFab(lay, lib.NODE, 1, ng); s = lib.MultiFab(lay, lib.CELL, 1, ng)
    x.set_from_global(node_field(Xg, ng), (-ng,) * 3); r.set_from_global(node_field(Rg, ng), (-ng,) * 3); s.set_from_global(cell_field(Sg, ng), (-ng,) * 3)
    N.nodal_gs_sweep(g, x, r, s, fused)
    x.fill_boundary(g)
    res.append(x.gather_valid(n)[..., 0])
d = np.abs(res[0] - res[1])
print("max", d.max(), "count", (d > 0).sum())
idx = np.argwhere(d > 0)
print(idx[:40])
