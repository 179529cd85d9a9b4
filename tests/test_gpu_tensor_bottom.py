"""GPU: the direct bottom solve of the tensor operator (round 5, mlmg.hip: bottom_direct_prepare / k_dense_bottom).  The coarsest level of
an MLTensorOp hierarchy (2^3 ... 3^3 cells, three coupled components) is handed to BiCGStab by amrex::MLMG (bottom_reltol 1e-4; the role of
the bottom solver inside MLMG::mgVcycle for the solves of Diffusion.cpp:804-957 and :1010-1178).  With constant viscosity the product
assembles the level's operator matrix (cached per distinct level) and solves it exactly in one single-workgroup launch.  A bottom solver
is one of the unpinned solver choices (tests/test_gpu_sensitivity.py): it may not change a converged answer beyond the solver tolerance.
Here: the same tensor solves under both bottom solvers (IAMRX_TENSOR_BOTTOM_DIRECT = 1 / 0) -- periodic, no-slip / slip walls with
per-component conditions, a domain that coarsens to 3^3 cells, variable density -- and the operator matrix itself against the operator."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DIRICHLET, NEUMANN = 101, 102


def solve(lib, N, n, per, lobc, hibc, bval, seed, rho_var, tol=1e-11):
    g = lib.Geom.make(n, periodic=per)
    lay = lib.Layout.single(n)
    rng = np.random.default_rng(seed)
    u = np.zeros(tuple(v + 2 for v in n) + (3,))
    u[1:-1, 1:-1, 1:-1, :] = rng.standard_normal(tuple(n) + (3,))
    eta_d = []
    for d in range(3):
        m = lib.MultiFab(lay, lib.face(d), 1, 0); m.setval(0.013); eta_d.append(m)
    a = 1.0 + (0.4 * rng.random(tuple(n) + (1,)) if rho_var else np.zeros(tuple(n) + (1,)))
    a_d = lib.MultiFab(lay, lib.CELL, 1, 0); a_d.set_from_global(a, (0, 0, 0))
    r_d = lib.MultiFab(lay, lib.CELL, 3, 0); r_d.set_from_global(u[1:-1, 1:-1, 1:-1, :], (0, 0, 0))
    s_d = lib.MultiFab(lay, lib.CELL, 3, 1); s_d.setval(0.0)
    kw = {} if lobc is None else dict(lobc=lobc, hibc=hibc)
    st = N.tensor_solve(g, s_d, r_d, 1.0, bval, a_d, eta_d, tol_rel=tol, tol_abs=0.0, opts=lib.mg_opts(maxorder=2), **kw)
    return st, s_d.gather_valid(n)


D, Nm = DIRICHLET, NEUMANN
CASES = {
    "periodic 32^3 (bottom 2^3)": ((32, 32, 32), (1, 1, 1), None, None, 0.6, False),
    "periodic 48 x 24 x 24 (bottom 6 x 3 x 3: BiCGStab stays)": ((48, 24, 24), (1, 1, 1), None, None, 0.6, False),
    "periodic 24^3 (bottom 3^3), variable density": ((24, 24, 24), (1, 1, 1), None, None, 0.4, True),
    "lid-driven cavity walls (no-slip everywhere)": ((32, 32, 32), (0, 0, 0), [[D] * 3] * 3, [[D] * 3] * 3, 0.5, False),
    "slip walls in x / y, no-slip in z, per component": ((32, 16, 16), (0, 0, 0), [[D if (d == 2 or d == c) else Nm for d in range(3)] for c in range(3)],
                                                       [[D if (d == 2 or d == c) else Nm for d in range(3)] for c in range(3)], 0.5, True),
    "walls in z only": ((16, 16, 32), (1, 1, 0), [[0, 0, D]] * 3, [[0, 0, D]] * 3, 0.7, False),
}


@pytest.mark.parametrize("case", list(CASES))
def test_direct_bottom_solve_gives_the_converged_answer_of_the_krylov_bottom(gpu, case):
    lib = gpu
    from iamr_amd import ns as N
    n, per, lobc, hibc, bval, rho_var = CASES[case]
    old = lib.tuning_get("TENSOR_BOTTOM_DIRECT", 1)
    out = {}
    try:
        for mode in (1, 0):
            lib.tuning_set("TENSOR_BOTTOM_DIRECT", mode)
            out[mode] = solve(lib, N, n, per, lobc, hibc, bval, 11, rho_var)
    finally:
        lib.tuning_set("TENSOR_BOTTOM_DIRECT", old)
    (st1, s1), (st0, s0) = out[1], out[0]
    assert st1.converged == 1 and st0.converged == 1
    assert st1.nlevels == st0.nlevels and st1.nlevels >= 3
    assert abs(st1.iters - st0.iters) <= 1, (st1.iters, st0.iters)
    # both are converged to 1e-11 of the right-hand side's norm: the solutions agree to the solver tolerance (x the condition of the operator)
    assert np.abs(s1 - s0).max() <= 1e-9 * np.abs(s0).max(), float(np.abs(s1 - s0).max())


def test_direct_bottom_solve_satisfies_the_equation(gpu):
    """the converged solution under the direct bottom solver satisfies (a - b div tau) u = rhs (the operator applied through iamrx_tensor_apply)"""
    lib = gpu
    from iamr_amd import ns as N
    n, per = (32, 32, 32), (1, 1, 1)
    g = lib.Geom.make(n, periodic=per)
    lay = lib.Layout.single(n)
    rng = np.random.default_rng(3)
    rhs = rng.standard_normal(tuple(n) + (3,))
    eta_d = []
    for d in range(3):
        m = lib.MultiFab(lay, lib.face(d), 1, 0); m.setval(0.02); eta_d.append(m)
    a_d = lib.MultiFab(lay, lib.CELL, 1, 0); a_d.setval(1.3)
    r_d = lib.MultiFab(lay, lib.CELL, 3, 0); r_d.set_from_global(rhs, (0, 0, 0))
    s_d = lib.MultiFab(lay, lib.CELL, 3, 1); s_d.setval(0.0)
    assert lib.tuning_get("TENSOR_BOTTOM_DIRECT", 1) == 1
    st = N.tensor_solve(g, s_d, r_d, 1.0, 0.8, a_d, eta_d, tol_rel=1e-11, tol_abs=0.0)
    assert st.converged == 1
    out_d = lib.MultiFab(lay, lib.CELL, 3, 0)
    N.tensor_apply(g, out_d, s_d, 1.0, 0.8, a_d, eta_d)
    got = out_d.gather_valid(n)
    assert np.abs(got - rhs).max() <= 1e-10 * np.abs(rhs).max(), float(np.abs(got - rhs).max())
