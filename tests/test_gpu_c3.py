"""BASELINE config C3: DoubleShearLayer, 2-level AMR (ref_ratio 2), regrid + SyncRegister path (Source/prob/prob_init.cpp:346-405,
Tutorials/DoubleShearLayer).  The library is three-dimensional: C3 runs as a z-uniform periodic SLAB of the 3-D
algorithm -- every quote of C3 in this repository names the slab.  Two tests:

* the slab against the oracle (which is given the product's grids, as tests/test_gpu_amr_regrid.py does): initial grids from the
  vorticity of the shear layers, post_init, coarse steps with a regrid at the start of every step, state / grad p / pressure on every
  level after every step.  The oracle runs the literal box-by-box SyncRegister (oracle/orc_syncreg.c), the product its union-based one.
* the full-size slab of config C3 (512 x 512 base, here x 32 cells in z, two levels, regridded every step) through the same path with
  size-independent properties: z-independence, w = 0, constant density, composite mass of the tracer, kinetic energy not growing, the
  refined level following the shear layers."""
import numpy as np
import pytest

import orc
from test_gpu_amr_step import _compare

pytestmark = pytest.mark.gpu
# IAMRX_TEST_LONG = 1: the longer variants the suite ran before its time budget was cut in round 5 (one more coarse step per case, the 64^2
# C3 slab parity, two coarse steps in front of the plotfile comparison) -- kept, not deleted
LONG = __import__("os").environ.get("IAMRX_TEST_LONG") == "1"

PROB_LO, PROB_HI_XY = -1.0, 1.0


def _dsl_state(X, Y, Z, tracer=True):
    """probtype 5, direction 1 (iamr_amd/probinit.py restates prob_init.cpp:346-405) + a z-uniform tracer column instead of the
    tutorial's sphere (a sphere is not z-uniform)"""
    from iamr_amd import probinit
    S = probinit.initial_state(dict(probtype=5, direction=1, interface_width=1.0, blob_center=(0.0, 0.0, 0.0), blob_radius=0.0, density_ic=1.0), X, Y, Z)
    if tracer:
        S[..., 4] = np.exp(-((X - 0.5) ** 2 + (Y - 0.1) ** 2) / 0.02)
    return S


def _set_level(lib, lev, lay, n, prob_lo, prob_hi):
    from iamr_amd import probinit
    m = lib.MultiFab(lay, lib.CELL, 5, 1)
    for li in range(m.nlocal()):
        lo, hi = m.fab_box(li)
        X, Y, Z = probinit.cell_centres(n, prob_lo, prob_hi, lo, hi)
        m.from_numpy(_dsl_state(X, Y, Z), li)
    lev.set_data(lev.S_NEW, m)


def _build(lib, n, nz, vort, kw, max_grid_size, blocking_factor=4):
    """two-level hierarchy of the slab: level 0 = n x n x nz, level 1 from the vorticity tags of the initial data (Amr::bldFineLevels)"""
    from iamr_amd.ns import ns_params
    from iamr_amd.amr import Amr
    hz = 0.5 * nz * (PROB_HI_XY - PROB_LO) / n                       # dz = dx
    prob_lo, prob_hi = (PROB_LO, PROB_LO, -hz), (PROB_HI_XY, PROB_HI_XY, hz)
    g0 = lib.Geom.make([n, n, nz], prob_lo=prob_lo, prob_hi=prob_hi, periodic=(1, 1, 1))
    lay0 = lib.Layout.decompose([n, n, nz], max(n, nz))
    amr = Amr(g0, [lay0], ns_params(**kw), lib.mg_opts())
    _set_level(lib, amr.levels[0], lay0, [n, n, nz], prob_lo, prob_hi)
    amr.set_regrid(max_level=1, regrid_int=1, rules=[dict(comp=-1, mode=2, value=[vort])], blocking_factor=blocking_factor, max_grid_size=max_grid_size,
                   n_error_buf=1, grid_eff=0.75)
    assert amr.regrid()
    assert amr.nlev == 2
    _set_level(lib, amr.levels[1], amr.layouts[1], [2 * n, 2 * n, 2 * nz], prob_lo, prob_hi)      # initData, not the interpolant
    return amr, prob_lo, prob_hi


@pytest.mark.parametrize("n", [32, 64] if LONG else [32])          # (64: 50 s of oracle time; the GPU suite keeps its budget for the 64^3 step parity of test_gpu_ns.py)
def test_double_shear_layer_slab_matches_the_oracle(gpu, n):
    """the slab (n x n x 4 base cells, C3's parameters: cfl 0.5, inviscid, periodic [-1,1]^2, regrid_int 1, ratio 2) for three coarse steps"""
    lib = gpu
    nz = 4
    kw = dict(cfl=0.5, visc_coef=0.0, init_iter=2, init_shrink=1.0)
    amr, prob_lo, prob_hi = _build(lib, n, nz, 8.0 if n == 32 else 12.0, kw, max_grid_size=32)
    fine = [(tuple(lo), tuple(hi)) for lo, hi in amr.layouts[1].boxes]
    assert len(fine) >= 2                                              # two shear layers -> at least two boxes, abutting ones among them
    og = orc.geom([n, n, nz], problo=prob_lo, probhi=prob_hi)
    oa = orc.OrcAmr(og, orc.ns_params(**kw), orc.mg_opts(), [[], [(list(lo), list(hi)) for lo, hi in fine]])
    for l in range(2):
        X, Y, Z = oa.cell_centres(l)
        oa.set_state(l, _dsl_state(X, Y, Z))
    amr.post_init()
    oa.post_init()
    _compare(amr, oa, 5e-8, "after post_init")
    orc.lib().orc_syncreg_last_diff.restype = __import__("ctypes").c_double
    orc.lib().orc_syncreg_last_diff(1)
    changed = 0
    for step in range(3):
        before = list(amr.layouts[1].boxes)
        dt = amr.coarse_step()
        after = list(amr.layouts[1].boxes)
        if after != before:
            changed += 1
            dto = oa.regrid_then_step([[(tuple(lo), tuple(hi)) for lo, hi in after]])
        else:
            dto = oa.step()
        assert abs(dt - dto) <= 1e-8 * dto, (step, dt, dto)
        _compare(amr, oa, 5e-8, f"after coarse step {step + 1}")
    # the oracle's two registers (literal box-by-box / single-valued) agreed on every MLsyncProject right-hand side of the run
    assert orc.lib().orc_syncreg_last_diff(1) < 1e-8
    for l in range(2):
        lev = amr.levels[l]
        S = lev.data(lev.S_NEW).gather_valid(oa.n(l))
        cov = oa.cov(l)
        assert np.abs(S[..., 2])[cov].max() < 1e-10                   # w = 0
        assert np.abs(S - S[:, :, :1, :])[cov].max() < 1e-9            # no z-dependence


def _composite(amr, lib, comp, n, nz, weight_fn=None):
    """sum over the composite grid (cells of level 0 not under level 1 + cells of level 1) of state component comp (or of fn(S)) x cell volume"""
    tot = 0.0
    boxes1 = amr.layouts[1].boxes if amr.nlev > 1 else []
    under = np.zeros((n, n, nz), bool)
    for lo, hi in boxes1:
        under[lo[0] // 2:hi[0] // 2 + 1, lo[1] // 2:hi[1] // 2 + 1, lo[2] // 2:hi[2] // 2 + 1] = True
    for l in range(amr.nlev):
        lev = amr.levels[l]
        S = lev.data(lev.S_NEW)
        vol = np.prod([2.0 / (n * 2 ** l)] * 3)
        for li in range(S.nlocal()):
            a, lo = S.to_numpy(li)
            blo, bhi, _ = lev.layout.local_box(li)
            v = a[1:-1, 1:-1, 1:-1, :]
            q = v[..., comp] if weight_fn is None else weight_fn(v)
            if l == 0:
                q = q * ~under[blo[0]:bhi[0] + 1, blo[1]:bhi[1] + 1, blo[2]:bhi[2] + 1]
            tot += q.sum() * vol
    return tot


def test_c3_full_size_slab_properties(gpu):
    """config C3 at its full base resolution, 512 x 512 (x 32 cells in z: the geometric multigrid coarsens all directions together, as
    amrex::MLMG does, so a slab of 8 cells would stop the hierarchy at 128 x 128 x 2 and leave the bottom solver a 32 k-cell problem),
    one refined level following the vorticity of the two shear layers, regridded at the start of every coarse step: size-independent
    properties of the state through three coarse steps"""
    lib = gpu
    n, nz = 512, 32
    # ns.do_cons_trac = 1: the tracer is carried as S = rho q and advected conservatively, so that its composite mass is an invariant
    # proj.proj_tol = 1e-10 (reference default 1e-12, Projection.cpp:26): on the 1024^2-equivalent fine level the divergence of the tanh
    # layers is O(10) and 1/h^2 = 2.6e5, the composite residual bottoms out at 2.7e-11 = 3e-12 relative in fp64 -- upstream's criterion
    # res <= rtol * max(|rhs|, |res0|) with 1e-12 is below that floor (amrex::MLMG would abort with "failed to converge" as well)
    kw = dict(cfl=0.5, visc_coef=0.0, init_iter=2, init_shrink=1.0, do_cons_trac=1, proj_tol=1.0e-10)
    amr, prob_lo, prob_hi = _build(lib, n, nz, 20.0, kw, max_grid_size=128, blocking_factor=8)
    amr.post_init()
    ke = lambda v: 0.5 * (v[..., 0] ** 2 + v[..., 1] ** 2 + v[..., 2] ** 2)
    m0, t0, e0 = _composite(amr, lib, 3, n, nz), _composite(amr, lib, 4, n, nz), _composite(amr, lib, 0, n, nz, ke)
    grids = [list(amr.layouts[1].boxes)]
    for step in range(3):
        amr.coarse_step()
        assert amr.nlev == 2
        grids.append(list(amr.layouts[1].boxes))
    m1, t1, e1 = _composite(amr, lib, 3, n, nz), _composite(amr, lib, 4, n, nz), _composite(amr, lib, 0, n, nz, ke)
    assert abs(m1 - m0) <= 1e-11 * m0                                  # density (conservative, constant)
    assert abs(t1 - t0) <= 1e-10 * abs(t0), (t0, t1)                   # conservative tracer: advection, reflux, average down and regrid keep it
    assert e1 <= e0 * (1 + 1e-7) and e1 >= 0.99 * e0, (e0, e1)         # inviscid Godunov + approximate projection: no energy growth beyond the projection's own O(1e-9) slack
    cov = np.zeros((2 * n, 2 * n), bool)
    for l in range(2):
        lev = amr.levels[l]
        S = lev.data(lev.S_NEW)
        for li in range(S.nlocal()):
            a, lo = S.to_numpy(li)
            v = a[1:-1, 1:-1, 1:-1, :]
            assert np.abs(v[..., 2]).max() < 1e-10                    # w = 0
            assert np.abs(v - v[:, :, :1, :]).max() < 1e-9             # no z-dependence
            assert np.abs(v[..., 3] - 1.0).max() < 1e-11               # constant density stays constant
            if l == 1:
                blo, bhi, _ = lev.layout.local_box(li)
                assert blo[2] == 0 and bhi[2] == 2 * nz - 1             # fine boxes span the slab
                cov[blo[0]:bhi[0] + 1, blo[1]:bhi[1] + 1] = True
    # the refined level covers the two shear layers |x| = 0.5 over the whole y-extent and leaves the quiet middle and edges to level 0
    for xs in (-0.5, 0.5):
        ic = int((xs - PROB_LO) / (PROB_HI_XY - PROB_LO) * 2 * n)
        assert cov[ic - 4:ic + 4, :].all()
    assert not cov[2 * n // 2 - 32:2 * n // 2 + 32, :].any() and not cov[:32, :].any()
    assert 0.02 < cov.mean() < 0.5


def test_c3_bench_workload_from_the_2d_inputs(gpu):
    """bench.py's C3 line at a small size: the committed 2-D inputs (tests/golden/inputs.2d.doubleshearlayer_c3) lifted onto its 8-cell
    slab with slab multigrid levels, two levels, regrid every step -- the run completes on two levels and reports the plane's cells"""
    import os, sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    r = bench.c3_workload(gpu, 128, 2)
    assert r["levels"] == 2 and r["fine_level_cells2d"] > 0 and 0.01 < r["fine_level_cover"] < 0.6, r
    assert r["cells2d_per_sec"] > 0 and gpu.tuning_get("MG_SLAB", 0) == 0
