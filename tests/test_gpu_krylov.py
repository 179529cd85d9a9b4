"""BiCGStab of the multigrid bottom solvers with its scalars on the device (iamr_amd/csrc/krylov.h, round 6) against the host-driven loop
it replaces (CellMG::bicgstab / NodalMG::bicgstab: amrex::MLCGSolver::solve_bicgstab with five read-backs per iteration).  Same operations
on the same doubles in the same order: iteration counts equal, solutions equal to round-off of one multiply-add (the compiler is free in how
it orders a kernel's independent loads, not in its arithmetic: the build uses -ffp-contract=off)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _field(n, ng, seed):
    rng = np.random.default_rng(seed)
    ax = [(np.arange(-ng, n[d] + ng) + 0.5) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    ph = rng.uniform(0, 2 * np.pi, 4)
    return np.sin(2 * np.pi * X + ph[0]) * np.cos(2 * np.pi * Y + ph[1]) + 0.5 * np.cos(4 * np.pi * Z + ph[2]) + 0.01 * rng.standard_normal(X.shape)


@pytest.mark.parametrize("boxes", [None, 8, 16])
@pytest.mark.parametrize("bc", ["periodic", "dirichlet"])
def test_cell_solve_same_iterates_as_host_driven_loop(gpu, boxes, bc):
    lib = gpu
    n = (32, 32, 32)
    per = (1, 1, 1) if bc == "periodic" else (0, 0, 0)
    lobc = hibc = (0, 0, 0) if bc == "periodic" else (101, 101, 101)
    g = lib.Geom.make(n, periodic=per)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    b = []
    for d in range(3):
        typ = tuple(int(q == d) for q in range(3))
        m = lib.MultiFab(lay, typ, 1, 0)
        shape = tuple(n[q] + typ[q] for q in range(3))
        ax = [(np.arange(shape[q]) + (0.0 if typ[q] else 0.5)) / n[q] for q in range(3)]
        X, Y, Z = np.meshgrid(*ax, indexing="ij")
        m.set_from_global((1.0 + 0.4 * np.sin(2 * np.pi * X + d) * np.cos(2 * np.pi * Y) * np.cos(2 * np.pi * Z))[..., None], (0, 0, 0))
        b.append(m)
    rhs = _field(n, 0, 11)
    rhs -= rhs.mean()
    rhs_d = lib.MultiFab(lay, lib.CELL, 1, 0); rhs_d.set_from_global(rhs[..., None], (0, 0, 0))
    out = {}
    for dev in (0, 1):
        lib.tuning_set("KRYLOV_DEVICE", dev)
        try:
            phi_d = lib.MultiFab(lay, lib.CELL, 1, 1); phi_d.setval(0.0)
            # min_width 8: the hierarchy ends at 8^3 (4^3 per box of the 8-box layout), so the Krylov solver does real work
            st = lib.abec_solve(g, 0.0, 1.0, None, b, phi_d, rhs_d, lobc, hibc, rtol=1e-11, atol=0.0,
                                opts=lib.mg_opts(device_bottom=0, min_width=8))
        finally:
            lib.tuning_set("KRYLOV_DEVICE", 1)
        assert st.converged == 1
        out[dev] = (st.iters, st.bottom_iters_total, phi_d.gather_valid(n)[..., 0])
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1], (out[0][:2], out[1][:2])
    assert out[1][1] > out[1][0], "the Krylov solver should have iterated"
    assert np.abs(out[0][2] - out[1][2]).max() <= 1e-13 * np.abs(out[0][2]).max()


def _run(lib, N, boxes, dev):
    n = (16,) * 3
    lib.tuning_set("KRYLOV_DEVICE", dev)
    try:
        g = lib.Geom.make(n)
        lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
        ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.5, visc_coef=1e-2, init_iter=2), lib.mg_opts(device_bottom=0))
        ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
        ns.post_init(-1.0)
        dts = [ns.step() for _ in range(2)]
        S = ns.data(N.NavierStokes.S_NEW).gather_valid(n)
        P = ns.data(N.NavierStokes.P_NEW).gather_valid(n)[..., 0]
    finally:
        lib.tuning_set("KRYLOV_DEVICE", 1)
    return dts, S, P


@pytest.mark.parametrize("boxes", [None, 8])
def test_time_steps_same_with_device_and_host_driven_krylov(gpu, boxes):
    """MAC, viscous and nodal solves of two TaylorGreen steps, every multigrid ending in BiCGStab on the 2^3 level (device_bottom = 0)"""
    lib = gpu
    from iamr_amd import ns as N
    d0, S0, P0 = _run(lib, N, boxes, 0)
    d1, S1, P1 = _run(lib, N, boxes, 1)
    assert d0 == d1
    assert np.abs(S0 - S1).max() <= 1e-13 and np.abs(P0 - P1).max() <= 1e-12
