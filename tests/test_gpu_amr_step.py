"""the multi-level time step (subcycled advance of a refined level, flux / sync registers, reflux, average down,
MAC sync, MLsyncProject, multi-level initialisation) -- HIP hierarchy (iamrx_amr_*, through the C-ABI) against the CPU oracle
(oracle/orc_amr.c), which solves the composite nodal systems with a different solver (conjugate gradients on the conforming
composite operator) and holds every refined level as a whole-domain array.

Tolerances: the composite sync solves stop at sync_tol = 1e-10 (relative), the level solves at 1e-12; the states of the two codes
agree to <= 2e-8 of the velocity scale after full coarse steps."""
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


def _make(n0, fine_boxes, crse_split, params_kw, state_fn):
    from iamr_amd import lib as L
    from iamr_amd.ns import ns_params
    from iamr_amd.amr import Amr
    L.init()
    g0 = L.Geom.make([n0] * 3)
    lay0 = L.Layout.decompose([n0] * 3, crse_split)
    lay1 = L.Layout([(tuple(lo), tuple(hi)) for lo, hi in fine_boxes])
    amr = Amr(g0, [lay0, lay1], ns_params(**params_kw), L.mg_opts())
    og = orc.geom([n0] * 3)
    oa = orc.OrcAmr(og, orc.ns_params(**params_kw), orc.mg_opts(), [[], fine_boxes])
    for l in range(2):
        S = state_fn(*oa.cell_centres(l))
        oa.set_state(l, S)
        lev = amr.levels[l]
        mf = L.MultiFab(lev.layout, L.CELL, 5, 1)
        G = np.zeros(tuple(s + 2 for s in S.shape[:3]) + (5,), order="F")
        G[1:-1, 1:-1, 1:-1] = S
        mf.set_from_global(G, (-1, -1, -1))
        lev.set_data(lev.S_NEW, mf)
    return amr, oa


def _compare(amr, oa, tol, tag):
    worst = 0.0
    for l in range(2):
        lev = amr.levels[l]
        n = oa.n(l)
        cov = oa.cov(l)
        S = lev.data(lev.S_NEW).gather_valid(n)
        So = oa.state(l)
        scale = max(1.0, np.abs(So[cov]).max())
        err = np.abs(S - So)[cov].max() / scale
        assert err <= tol, f"{tag}: level {l} state differs by {err}"
        worst = max(worst, err)
        Gp = lev.data(lev.GP_NEW).gather_valid(n)
        Go = oa.fab(l, 4).valid(n)
        gs = max(1.0, np.abs(Go[cov]).max())
        gerr = np.abs(Gp - Go)[cov].max() / gs
        assert gerr <= 50 * tol, f"{tag}: level {l} grad p differs by {gerr}"
        # pressure on the nodes of the level's cells
        P = lev.data(lev.P_NEW).gather_valid(n)[..., 0]
        Po = oa.fab(l, 2).valid(n, orc.NODE)[..., 0]
        nm = np.zeros(P.shape, bool)
        for dz in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    nm[dx:dx + n[0], dy:dy + n[1], dz:dz + n[2]] |= cov
        d = (P - Po)[nm]
        if oa.g0.periodic[0] and oa.g0.periodic[1] and oa.g0.periodic[2]:
            pass          # the composite systems fix the additive constant the same way (mean-free right-hand sides)
        perr = np.abs(d).max() / max(1.0, np.abs(Po[nm]).max())
        assert perr <= 100 * tol, f"{tag}: level {l} pressure differs by {perr}"
    return worst


def _composite_sum(amr_levels_state, cov1, dx0, dx1, comp):
    S0, S1 = amr_levels_state
    covc = cov1[::2, ::2, ::2]
    return (S0[..., comp] * ~covc).sum() * np.prod(dx0) + (S1[..., comp] * cov1).sum() * np.prod(dx1)


@pytest.mark.parametrize("case", ["one_box", "l_shape"])
def test_two_level_taylorgreen_matches_oracle(case):
    n0 = 16
    if case == "one_box":
        fine = [([4, 4, 4], [19, 19, 19])]
        split = 16
    else:
        # L-shaped refined region made of three boxes (the shape of Exec/run2d/test_grids/fixed_grids_2), coarse level in 8 boxes
        fine = [([8, 8, 8], [15, 15, 23]), ([16, 8, 8], [23, 15, 23]), ([8, 16, 8], [15, 23, 23])]
        split = 8
    kw = dict(cfl=0.7, visc_coef=0.0, init_iter=2)
    amr, oa = _make(n0, fine, split, kw, lambda X, Y, Z: orc.taylorgreen_state(X, Y, Z, c=1.0))
    amr.post_init()
    oa.post_init()
    assert abs(amr.dts()[0] - oa.dt(0)) <= 1e-9 * oa.dt(0) and abs(amr.dts()[1] - oa.dt(1)) <= 1e-9 * oa.dt(1)
    _compare(amr, oa, 2e-8, "after post_init")
    cov1 = oa.cov(1)
    m0 = [_composite_sum((oa.state(0), oa.state(1)), cov1, oa.dx(0), oa.dx(1), c) for c in range(5)]
    for step in range(2):
        dt = amr.coarse_step()
        dto = oa.step()
        assert abs(dt - dto) <= 1e-8 * dto
        _compare(amr, oa, 2e-8, f"after coarse step {step + 1}")
        st, stm = amr.sync_stats()
        assert st.converged == 1 and stm.converged == 1
    # composite conservation of mass, tracer and momentum (periodic domain, conservative updates + reflux + sync)
    n = [oa.n(0), oa.n(1)]
    Sg = [amr.levels[l].data(0).gather_valid(n[l]) for l in range(2)]
    m1 = [_composite_sum(Sg, cov1, oa.dx(0), oa.dx(1), c) for c in range(5)]
    assert abs(m1[3] - m0[3]) <= 1e-12
    assert abs(m1[4] - m0[4]) <= 1e-11
