"""nodal projection V-cycle cost under different options (scratch tool)"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from iamr_amd import lib
from iamr_amd import ns as N
lib.init(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = lib.Geom.make((n,)*3); lay = lib.Layout.single((n,)*3)
vel = lib.MultiFab(lay, lib.CELL, 3, 1)
x = (np.arange(-1, n+1)+0.5)/n
X, Y, Z = np.meshgrid(x, x, x, indexing='ij')
a = np.zeros((n+2,)*3+(3,), order='F')
a[...,0] = np.sin(2*np.pi*X)*np.cos(2*np.pi*Y)+0.3*np.cos(2*np.pi*Z)
a[...,1] = -np.cos(2*np.pi*X)*np.sin(2*np.pi*Y)+0.2*np.sin(4*np.pi*Z)
a[...,2] = 0.5*np.sin(2*np.pi*Z)*np.cos(2*np.pi*X)
sig = lib.MultiFab(lay, lib.CELL, 1, 1); sig.setval(1.0)
cfgs = [("default", {}), ("default2", {})]
for ns_ in (1, 2, 3, 4):
    for nu in ((1, 1), (2, 2), (2, 1), (1, 2)):
        cfgs.append((f"sweeps{ns_} nu{nu[0]}{nu[1]}", {"nodal_sweeps": ns_, "nu1": nu[0], "nu2": nu[1]}))
for label, kw in cfgs:
    vel.from_numpy(a)
    phi = lib.MultiFab(lay, lib.NODE, 1, 1); phi.setval(0.0)
    lib.sync(); t0 = time.perf_counter()
    st = N.nodal_projection(g, vel, 0, phi, sig, rel_tol=1e-11, opts=lib.mg_opts(**kw))
    lib.sync(); wall = (time.perf_counter()-t0)*1e3
    print(f"{label:16s} iters {st.iters} vcycle_ms {st.vcycle_ms:.2f} wall {wall:.1f} bottom_its {st.bottom_iters_total} res {st.resnorm:.2e}", flush=True)
