"""scratch: does the nodal projection on a slab hierarchy (tests/test_gpu_slab_mg.py, walls) depend on what the allocator hands out?"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from iamr_amd import lib, ns as N
from test_gpu_slab_mg import plane_fields, NEUMANN, PERIODIC
lib.init(0)
n = (128, 8, 64)
per = (0, 1, 1); lobc = (NEUMANN, PERIODIC, PERIODIC)
g = lib.Geom.make(n, prob_hi=tuple(v / n[0] for v in n), periodic=per)
lay = lib.Layout.single(n)
rho, _ = plane_fields(n, 5, True)
x = (np.arange(-1, n[0] + 1) + 0.5) / n[0]; z = (np.arange(-1, n[2] + 1) + 0.5) / n[2]
X, Z = np.meshgrid(x, z, indexing="ij")
u2 = np.sin(np.pi * X) * np.cos(2 * np.pi * Z); w2 = np.cos(2 * np.pi * X) * np.sin(4 * np.pi * Z)
vel = np.zeros(tuple(v + 2 for v in n) + (3,))
vel[..., 0] = u2[:, None, :]; vel[..., 2] = w2[:, None, :]
vel[0, ..., 0] = -vel[1, ..., 0]; vel[-1, ..., 0] = -vel[-2, ..., 0]


def run(mode, **okw):
    lib.tuning_set("MG_SLAB", mode)
    sig_d = lib.MultiFab(lay, lib.CELL, 1, 1); sig_d.set_from_global((1.0 / rho)[..., None], (-1,) * 3)
    vel_d = lib.MultiFab(lay, lib.CELL, 3, 1); vel_d.set_from_global(vel, (-1,) * 3)
    p_d = lib.MultiFab(lay, lib.NODE, 1, 1); p_d.setval(0.0)
    st = N.nodal_projection(g, vel_d, 0, p_d, sig_d, 0, lobc=lobc, hibc=lobc, rel_tol=1e-11, abs_tol=1e-16, opts=lib.mg_opts(**okw))
    v = vel_d.gather_valid(n)
    return st.iters, st.converged, st.nlevels, float(np.abs(v[..., 1]).max()), st.resnorm, float(np.abs(p_d.gather_valid(n)).sum())


def dirty(val):
    ms = []
    for nn in ((128, 8, 64), (64, 4, 32), (32, 2, 16), (16, 2, 8), (8, 2, 4), (130, 10, 66)):
        for t in (lib.CELL, lib.NODE):
            for ng in (0, 1):
                m = lib.MultiFab(lib.Layout.single(nn), t, 3, ng); m.setval(val); ms.append(m)
    del ms






for kw in (dict(max_coarsening_level=2), dict(max_coarsening_level=3), dict(max_coarsening_level=3, device_bottom=0),            dict(max_coarsening_level=4), dict(max_coarsening_level=3, fixed_iters=1), dict(max_coarsening_level=2, fixed_iters=1)):
    try:
        out = [run(1, **kw) for _ in range(5)]
    except Exception as e:
        print(kw, 'FAILED', e); continue
    print(kw, "deterministic" if all(o == out[0] for o in out) else "VARIES", [o[2] for o in out][:1], [o[0] for o in out], ["%.4e" % o[4] for o in out])
