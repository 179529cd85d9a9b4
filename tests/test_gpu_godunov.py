"""GPU parity: Godunov PLM kernels (ExtrapVelToFaces, ComputeAofs chain) through the C-ABI against the
CPU oracle on identical seeded inputs.  Both sides are built with -ffp-contract=off and evaluate the
same expressions in the same order, so the bar is bit-exact."""
import ctypes as C
import numpy as np
import pytest
from conftest import godunov_same

pytestmark = pytest.mark.gpu


def field(n, ng, seed, amp=1.0, typ=(0, 0, 0)):
    rng = np.random.default_rng(seed)
    ax = [(np.arange(-ng, n[d] + typ[d] + ng) + (0.0 if typ[d] else 0.5)) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    ph = rng.uniform(0, 2 * np.pi, 6)
    f = (np.sin(2 * np.pi * X + ph[0]) * np.cos(2 * np.pi * Y + ph[1]) + 0.5 * np.cos(4 * np.pi * Z + ph[2]) * np.sin(2 * np.pi * X + ph[3])
         + 0.25 * np.sin(2 * np.pi * (Y + Z) + ph[4]))
    return amp * f


def periodic_fab(orc, L, g, n, typ, ng, nc, seeds, amp=1.0):
    f = orc.Fab(n, typ, ng, nc)
    for c in range(nc):
        f.a[..., c] = field(n, ng, seeds + c, amp, typ)
    L.orc_fill_periodic(f.ref(), C.byref(g), orc.i3(typ))
    return f


def to_dev(lib, lay, f, typ, ng):
    m = lib.MultiFab(lay, typ, f.nc, ng)
    m.set_from_global(f.a, f.lo)
    return m


@pytest.mark.parametrize("n,boxes,fit", [((16, 16, 16), None, 0), ((32, 16, 24), None, 1), ((32, 32, 32), 16, 0)])
def test_extrap_vel_to_faces(orc, gpu, n, boxes, fit):
    lib = gpu
    L = orc.lib()
    g_o = orc.geom(n)
    g_d = lib.Geom.make(n)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    vel = periodic_fab(orc, L, g_o, n, orc.CELL, 3, 3, 100)
    # include exact zeros / tiny values to exercise the small_vel branches
    vel.a[np.abs(vel.a) < 0.02] = 0.0
    L.orc_fill_periodic(vel.ref(), C.byref(g_o), orc.i3(orc.CELL))
    force = periodic_fab(orc, L, g_o, n, orc.CELL, 1, 3, 200, amp=3.0)
    um_o = [orc.Fab(n, orc.face(d), 1, 1) for d in range(3)]
    dt = 0.4 / max(n)
    L.orc_extrap_vel_to_faces(C.byref(g_o), vel.ref(), force.ref(), orc.fabptrs(um_o), C.c_double(dt), orc.bcrecs(3), fit)
    vel_d = to_dev(lib, lay, vel, lib.CELL, 3)
    force_d = to_dev(lib, lay, force, lib.CELL, 1)
    um_d = [lib.MultiFab(lay, lib.face(d), 1, 1) for d in range(3)]
    lib.godunov_extrap_vel_to_faces(g_d, vel_d, force_d, um_d, dt, None, fit)
    for d in range(3):
        got = um_d[d].gather_valid(n)[..., 0]
        ref = um_o[d].valid(n, orc.face(d))[..., 0]
        godunov_same(got, ref, d)


@pytest.mark.parametrize("n,boxes,ncomp,iconserv,isvel,fit", [
    ((16, 16, 16), None, 3, (0, 0, 0), 1, 0),
    ((32, 16, 24), None, 2, (1, 0), 0, 0),
    ((16, 16, 16), None, 2, (1, 1), 0, 1),
    ((32, 32, 32), 16, 3, (0, 0, 0), 1, 0),
])
def test_compute_aofs(orc, gpu, n, boxes, ncomp, iconserv, isvel, fit):
    lib = gpu
    L = orc.lib()
    g_o = orc.geom(n)
    g_d = lib.Geom.make(n)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    S = periodic_fab(orc, L, g_o, n, orc.CELL, 3, ncomp, 300)
    force = periodic_fab(orc, L, g_o, n, orc.CELL, 1, ncomp, 400, amp=2.0)
    divu = periodic_fab(orc, L, g_o, n, orc.CELL, 1, 1, 500, amp=0.3)
    um_o = []
    for d in range(3):
        t = orc.face(d)
        f = orc.Fab(n, t, 1, 1)
        f.a[..., 0] = field(n, 1, 600 + d, 1.0, t)
        f.a[np.abs(f.a) < 0.02] = 0.0
        # periodic duplicate face must agree
        hi = [slice(None)] * 3
        lo = [slice(None)] * 3
        hi[d] = 1 + n[d]
        lo[d] = 1
        f.a[tuple(hi)] = f.a[tuple(lo)]
        L.orc_fill_periodic(f.ref(), C.byref(g_o), orc.i3(t))
        um_o.append(f)
    aofs_o = orc.Fab(n, orc.CELL, 0, 5)
    edge_o = [orc.Fab(n, orc.face(d), 0, ncomp) for d in range(3)]
    ic = (C.c_int * ncomp)(*iconserv)
    dt = 0.4 / max(n)
    L.orc_compute_aofs(C.byref(g_o), aofs_o.ref(), 1, S.ref(), ncomp, force.ref(), divu.ref(), orc.fabptrs(um_o), ic,
                       C.c_double(dt), orc.bcrecs(ncomp), isvel, fit, orc.fabptrs(edge_o), None)
    S_d = to_dev(lib, lay, S, lib.CELL, 3)
    force_d = to_dev(lib, lay, force, lib.CELL, 1)
    divu_d = to_dev(lib, lay, divu, lib.CELL, 1)
    um_d = [to_dev(lib, lay, um_o[d], lib.face(d), 1) for d in range(3)]
    aofs_d = lib.MultiFab(lay, lib.CELL, 5, 0)
    aofs_d.setval(0.0)
    edge_d = [lib.MultiFab(lay, lib.face(d), ncomp, 0) for d in range(3)]
    lib.godunov_compute_aofs(g_d, aofs_d, 1, S_d, ncomp, force_d, divu_d, um_d, iconserv, dt, None, isvel, fit, edge=edge_d)
    for d in range(3):
        got = edge_d[d].gather_valid(n)
        ref = edge_o[d].valid(n, orc.face(d))
        godunov_same(got, ref, ("edge", d))
    got = aofs_d.gather_valid(n)[..., 1:1 + ncomp]
    ref = aofs_o.valid(n)[..., 1:1 + ncomp]
    godunov_same(got, ref, "aofs")
