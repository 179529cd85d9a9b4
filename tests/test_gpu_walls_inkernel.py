"""GPU: domain walls applied INSIDE the colour-pass kernels (WallK, k_abec.hip; GSRB_WALLS_INKERNEL) -- on a level that is one box spanning
a domain with walls, k_abec_gsrb / k_abec_gsrb1 form the homogeneous ghost value of k_abec_bc from the values they hold, and
CellMG::smooth launches no boundary fill in front of a colour pass.  The formula is the fill's (same operands, same order), so the solves
must agree BIT FOR BIT with the filled form (GSRB_WALLS_INKERNEL = 0), which tests/test_gpu_walls.py, test_gpu_ldc.py and test_gpu_abec.py
compare with the oracle: cell-centred MAC solve with Neumann / mixed periodic walls (coefficient arrays on the coarse levels:
k_abec_gsrb1<0>), scalar diffusion-type solve with Dirichlet walls of order 2 and 3, and the three-component tensor solve with one boundary
condition set per component (k_abec_gsrb<true, 2> / <true, 0>)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PERIODIC, DIRICHLET, NEUMANN = 0, 101, 102


@pytest.fixture()
def wallk(gpu):
    old = gpu.tuning_get("GSRB_WALLS_INKERNEL", 1)
    yield gpu
    gpu.tuning_set("GSRB_WALLS_INKERNEL", old)


def both(lib, fn):
    out = {}
    for mode in (0, 1):
        lib.tuning_set("GSRB_WALLS_INKERNEL", mode)
        out[mode] = fn()
    return out


@pytest.mark.parametrize("n,per", [((64, 32, 48), (0, 0, 0)), ((48, 32, 32), (1, 0, 0)), ((32, 32, 64), (0, 1, 0))])
def test_mac_solve_with_walls(wallk, n, per):
    lib = wallk
    g = lib.Geom.make(n, periodic=per)
    lay = lib.Layout.single(n)
    bc = tuple(PERIODIC if p else NEUMANN for p in per)
    rng = np.random.default_rng(11)
    ax = [(np.arange(-1, n[d] + 1) + 0.5) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    rho = 1.0 + 0.4 * np.cos(2 * np.pi * X) * np.cos(2 * np.pi * Y) + 0.2 * np.cos(4 * np.pi * Z)
    S = rng.standard_normal(n)
    S -= S.mean()

    def run():
        rho_d = lib.MultiFab(lay, lib.CELL, 1, 1); rho_d.set_from_global(rho[..., None], (-1,) * 3)
        S_d = lib.MultiFab(lay, lib.CELL, 1, 0); S_d.set_from_global(S[..., None], (0,) * 3)
        um = [lib.MultiFab(lay, lib.face(d), 1, 0) for d in range(3)]
        for m in um:
            m.setval(0.0)
        phi = lib.MultiFab(lay, lib.CELL, 1, 1); phi.setval(0.0)
        st = lib.mlmg_mac_solve(g, um, rho_d, 0, S_d, phi, 1.0, bc, bc, 1e-10, 1e-16)
        return st.iters, phi.gather_valid(n)

    r = both(lib, run)
    assert r[0][0] == r[1][0] and r[0][0] > 2
    assert np.array_equal(r[0][1], r[1][1])


@pytest.mark.parametrize("maxorder", [2, 3])
@pytest.mark.parametrize("uniform", [True, False])
def test_tensor_solve_with_per_component_walls(wallk, uniform, maxorder):
    lib = wallk
    from iamr_amd import ns as N
    n = (32, 24, 16)
    g = lib.Geom.make(n, periodic=(0, 0, 0))
    lay = lib.Layout.single(n)
    D, Nm = DIRICHLET, NEUMANN
    lobc = [[D if (d == 2 or d == c) else Nm for d in range(3)] for c in range(3)]      # x-lo / y-lo slip, everything else no-slip
    hibc = [[D, D, D] for c in range(3)]
    rng = np.random.default_rng(5)
    u = np.zeros(tuple(v + 2 for v in n) + (3,))
    u[1:-1, 1:-1, 1:-1, :] = rng.standard_normal(tuple(n) + (3,))
    u[:, :, -1, 0] = 1.0
    eta = [0.01 * (1.0 + (0.0 if uniform else 0.5) * rng.random(tuple(n[e] + (1 if e == d else 0) for e in range(3)) + (1,))) for d in range(3)]

    def run():
        eta_d = []
        for d in range(3):
            m = lib.MultiFab(lay, lib.face(d), 1, 0); m.set_from_global(eta[d], (0, 0, 0)); eta_d.append(m)
        a_d = lib.MultiFab(lay, lib.CELL, 1, 0); a_d.setval(1.0)
        r_d = lib.MultiFab(lay, lib.CELL, 3, 0); r_d.set_from_global(u[1:-1, 1:-1, 1:-1, :], (0, 0, 0))
        s_d = lib.MultiFab(lay, lib.CELL, 3, 1); s_d.set_from_global(u, (-1, -1, -1))
        st = N.tensor_solve(g, s_d, r_d, 1.0, 0.05, a_d, eta_d, lobc=lobc, hibc=hibc, tol_rel=1e-10, tol_abs=0.0, opts=lib.mg_opts(maxorder=maxorder))
        return st.iters, st.converged, s_d.gather_valid(n)

    r = both(lib, run)
    assert r[1][1] == 1 and r[0][0] == r[1][0]
    assert np.array_equal(r[0][2], r[1][2])
