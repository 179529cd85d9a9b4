"""GPU: the fused z-marching Godunov kernels (k_god_z / k_pred_z, iamr_amd/csrc/k_godunov.hip -- the default path of both Godunov_PLM and
Godunov_PPM) against the
multi-pass kernels (k_trace / k_dir / k_aofs, tuning key GODUNOV_Z = 0) through the C-ABI on the same device data: edge states, fluxes, aofs
and predicted face velocities agree to 1e-13 (FMA contraction may differ between the two instruction streams; both are compared with
the oracle in tests/test_gpu_godunov.py and tests/test_gpu_walls.py).  Sizes are chosen so that a box holds partial tiles,
several z-chunks and, in the wall cases, every BC branch; the periodic cases run the BC-free specialisation.  Both workgroup shapes of the
fused kernels run: 14 x 14 cells (the default: a grown tile of 16 x 16 = 256 threads) and 16 x 8 (GODUNOV_ZTX / GODUNOV_PTX = 16)."""
import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REFLECT_ODD, INT_DIR, REFLECT_EVEN, FOEXTRAP, EXT_DIR, HOEXTRAP = -1, 0, 1, 2, 3, 4
PER_BC5 = [((INT_DIR,) * 3, (INT_DIR,) * 3)] * 5
# no-slip walls in x, periodic y, slip walls in z (as tests/test_gpu_walls.py) + density / tracer rules
WALL_BC5 = [((EXT_DIR, INT_DIR, HOEXTRAP), (EXT_DIR, INT_DIR, HOEXTRAP)),
            ((EXT_DIR, INT_DIR, HOEXTRAP), (EXT_DIR, INT_DIR, HOEXTRAP)),
            ((EXT_DIR, INT_DIR, EXT_DIR), (EXT_DIR, INT_DIR, EXT_DIR)),
            ((FOEXTRAP, INT_DIR, FOEXTRAP), (FOEXTRAP, INT_DIR, REFLECT_EVEN)),
            ((FOEXTRAP, INT_DIR, FOEXTRAP), (HOEXTRAP, INT_DIR, REFLECT_ODD))]


def smooth(n, ng, seed, typ=(0, 0, 0)):
    rng = np.random.default_rng(seed)
    ax = [(np.arange(-ng, n[d] + typ[d] + ng) + (0.0 if typ[d] else 0.5)) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    ph = rng.uniform(0, 2 * np.pi, 5)
    f = (np.sin(2 * np.pi * X + ph[0]) * np.cos(2 * np.pi * Y + ph[1]) + 0.5 * np.cos(4 * np.pi * Z + ph[2]) * np.sin(2 * np.pi * X + ph[3])
         + 0.25 * np.sin(2 * np.pi * (Y + Z) + ph[4]))
    f[np.abs(f) < 0.02] = 0.0          # exact zeros: the small_vel branches
    return f


class path:
    def __init__(self, z, tile=14):
        self.kv = {"GODUNOV_Z": z, "GODUNOV_ZTX": tile, "GODUNOV_PTX": tile}

    def __enter__(self):
        from iamr_amd import lib
        self.old = {k: lib.tuning_get(k, 14 if k != "GODUNOV_Z" else 1) for k in self.kv}
        for k, v in self.kv.items():
            lib.tuning_set(k, v)

    def __exit__(self, *a):
        from iamr_amd import lib
        for k, v in self.old.items():
            lib.tuning_set(k, v)


def close(a, b, tag):
    err = float(np.abs(a - b).max())
    assert err <= 1e-13 * max(1.0, float(np.abs(b).max())), (tag, err)


@pytest.mark.parametrize("tile", [14, 16], ids=["14x14", "16x8"])
@pytest.mark.parametrize("scheme", [0, 1], ids=["plm", "ppm"])
@pytest.mark.parametrize("n,boxes,periodic,fit", [
    ((48, 40, 72), None, (1, 1, 1), 0),
    ((48, 40, 72), 24, (1, 1, 1), 1),
    ((40, 24, 48), None, (0, 1, 0), 0),
    ((32, 32, 32), 16, (0, 1, 0), 1),
])
def test_fused_equals_multipass(gpu, n, boxes, periodic, fit, scheme, tile):
    lib = gpu
    g = lib.Geom.make(n, periodic=periodic)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    bc5 = PER_BC5 if all(periodic) else WALL_BC5
    S = lib.MultiFab(lay, lib.CELL, 5, 3)
    G = np.stack([(1.5 if c >= 3 else 0.0) + smooth(n, 3, 10 + c) for c in range(5)], axis=-1)
    S.set_from_global(G, (-3, -3, -3))
    S.fill_boundary(g)
    if not all(periodic):
        S.fill_physbc(g, bc5, [[0.1 * (c + 1)] * 3 for c in range(5)], [[-0.05 * (c + 1)] * 3 for c in range(5)])
    frc = lib.MultiFab(lay, lib.CELL, 5, 1)
    frc.set_from_global(np.stack([2.0 * smooth(n, 1, 30 + c) for c in range(5)], axis=-1), (-1, -1, -1))
    divu = lib.MultiFab(lay, lib.CELL, 1, 1)
    divu.set_from_global(0.3 * smooth(n, 1, 50)[..., None], (-1, -1, -1))
    dt = 0.4 / max(n)
    # prediction
    um = {}
    for z in (0, 1):
        um[z] = [lib.MultiFab(lay, lib.face(d), 1, 1) for d in range(3)]
        for m in um[z]:
            m.setval(0.0)
        with path(z, tile):
            lib.godunov_extrap_vel_to_faces(g, S, frc, um[z], dt, bc5[:3], fit, scheme=scheme)
    for d in range(3):
        close(um[1][d].gather_valid(n), um[0][d].gather_valid(n), ("umac", d))
    # advection of all five components with mac velocities that carry ghost faces
    mac = []
    for d in range(3):
        t = lib.face(d)
        m = lib.MultiFab(lay, t, 1, 1)
        m.set_from_global(smooth(n, 1, 70 + d, t)[..., None], (-1, -1, -1))
        mac.append(m)
    out = {}
    for z in (0, 1):
        aofs = lib.MultiFab(lay, lib.CELL, 6, 0)
        aofs.setval(-7.0)
        edge = [lib.MultiFab(lay, lib.face(d), 5, 0) for d in range(3)]
        flux = [lib.MultiFab(lay, lib.face(d), 5, 0) for d in range(3)]
        with path(z, tile):
            lib.godunov_compute_aofs(g, aofs, 1, S, 5, frc, divu, mac, (0, 0, 0, 1, 0), dt, bc5, 1, fit, edge=edge, flux=flux, scheme=scheme)
        out[z] = (aofs.gather_valid(n), [e.gather_valid(n) for e in edge], [f.gather_valid(n) for f in flux])
    assert np.all(out[1][0][..., 0] == -7.0)                   # acomp offset respected
    close(out[1][0], out[0][0], "aofs")
    for d in range(3):
        close(out[1][1][d], out[0][1][d], ("edge", d))
        close(out[1][2][d], out[0][2][d], ("flux", d))


@pytest.mark.parametrize("n,boxes", [((72, 40, 48), None), ((64, 32, 64), 32)])
def test_split_bc_launches_equal_one_launch(gpu, n, boxes):
    """domain with walls: the tiles no boundary condition reaches run the plain code in a launch of their own (GODUNOV_SPLIT_BC = 1, thin
    first / last z-chunks), the boundary-condition variant takes the rest -- the one launch of the variant over all tiles to 1e-13 (two
    instantiations of the kernel: FMA contraction differs between their instruction streams; bit for bit in a STRICT_FP=1 build)"""
    lib = gpu
    periodic = (0, 1, 0)
    g = lib.Geom.make(n, periodic=periodic)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    S = lib.MultiFab(lay, lib.CELL, 5, 3)
    S.set_from_global(np.stack([(1.5 if c >= 3 else 0.0) + smooth(n, 3, 10 + c) for c in range(5)], axis=-1), (-3, -3, -3))
    S.fill_boundary(g)
    S.fill_physbc(g, WALL_BC5, [[0.1 * (c + 1)] * 3 for c in range(5)], [[-0.05 * (c + 1)] * 3 for c in range(5)])
    frc = lib.MultiFab(lay, lib.CELL, 5, 1)
    frc.set_from_global(np.stack([2.0 * smooth(n, 1, 30 + c) for c in range(5)], axis=-1), (-1, -1, -1))
    divu = lib.MultiFab(lay, lib.CELL, 1, 1)
    divu.set_from_global(0.3 * smooth(n, 1, 50)[..., None], (-1, -1, -1))
    mac = []
    for d in range(3):
        m = lib.MultiFab(lay, lib.face(d), 1, 1)
        m.set_from_global(smooth(n, 1, 70 + d, lib.face(d))[..., None], (-1, -1, -1))
        mac.append(m)
    dt = 0.4 / max(n)
    res = {}
    old = lib.tuning_get("GODUNOV_SPLIT_BC", 1)
    try:
        for split in (0, 1):
            lib.tuning_set("GODUNOV_SPLIT_BC", split)
            um = [lib.MultiFab(lay, lib.face(d), 1, 1) for d in range(3)]
            for m in um:
                m.setval(0.0)
            lib.godunov_extrap_vel_to_faces(g, S, frc, um, dt, WALL_BC5[:3], 1, scheme=0)
            aofs = lib.MultiFab(lay, lib.CELL, 5, 0)
            edge = [lib.MultiFab(lay, lib.face(d), 5, 0) for d in range(3)]
            lib.godunov_compute_aofs(g, aofs, 0, S, 5, frc, divu, mac, (0, 0, 0, 1, 0), dt, WALL_BC5, 1, 1, edge=edge, scheme=0)
            res[split] = [m.gather_valid(n) for m in um] + [aofs.gather_valid(n)] + [e.gather_valid(n) for e in edge]
    finally:
        lib.tuning_set("GODUNOV_SPLIT_BC", old)
    strict = os.environ.get("IAMRX_STRICT_FP") == "1"
    for i, (a, b) in enumerate(zip(res[0], res[1])):
        if strict:
            assert np.array_equal(a, b), i
        else:
            close(b, a, ("split", i))
