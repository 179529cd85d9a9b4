"""GPU, BASELINE.json's full single-GPU size (configs[1]: TaylorGreen 256^3 on one MI355X -- the workload bench.py times), where the
CPU oracle would take minutes per step: checked through size-independent properties instead.
  * known answer: with prob.c = 0 the TaylorGreen data is the 2-D Taylor vortex, an exact Navier-Stokes solution
    (reference Exec/benchmarks/EXACT_3D.F:75-119), u = sin(2 pi x) cos(2 pi y) exp(-8 pi^2 nu t);
  * invariants: no z-velocity and no z-dependence develop, mass and mean momentum are conserved;
  * decomposition independence: 8 boxes of 128^3 give the single-box answer to solver tolerance;
  * symmetries of the 3-D TaylorGreen flow (prob.c = 1, the bench configuration): x <-> y swap combined with a half-period shift,
    reflection in x.
All runs go through the C-ABI (iamrx_ns_*)."""
import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.boxes_kept]     # the 8-GPU box layouts on one GPU: boxes kept (tests/conftest.py)
N = 256


@pytest.fixture(scope="module")
def gpu():
    from iamr_amd import lib
    lib.init(0)
    return lib


def run(lib, mgs, nsteps, c, **kw):
    from iamr_amd import ns as NS
    n = (N,) * 3
    g = lib.Geom.make(n)
    lay = lib.Layout.single(n) if mgs is None else lib.Layout.decompose(n, mgs)
    ns = NS.NavierStokes(g, lay, NS.ns_params(init_iter=2, **kw))
    ns.init_taylorgreen(1.0, 1.0, 1.0, c, 1.0)
    ns.post_init(-1.0)
    for _ in range(nsteps):
        ns.step()
    S = ns.data(NS.NavierStokes.S_NEW).gather_valid(n)
    return S, ns.time


def test_taylor_vortex_known_answer_and_invariants_at_256(gpu):
    visc, dt, nsteps = 1.0e-2, 2.5e-3, 4
    S, t = run(gpu, None, nsteps, 0.0, cfl=0.7, visc_coef=visc, fixed_dt=dt)
    assert abs(t - nsteps * dt) < 1e-14
    x = (np.arange(N) + 0.5) / N
    dec = np.exp(-8 * np.pi ** 2 * visc * t)
    ue = (np.sin(2 * np.pi * x)[:, None] * np.cos(2 * np.pi * x)[None, :] * dec)[:, :, None]
    ve = (-np.cos(2 * np.pi * x)[:, None] * np.sin(2 * np.pi * x)[None, :] * dec)[:, :, None]
    err = max(np.abs(S[..., 0] - ue).max(), np.abs(S[..., 1] - ve).max())
    assert err < 2.0e-5, err                                    # second-order scheme at h = 1/256 (8^3 .. 16^3: oracle test, order 2)
    assert np.abs(S[..., 2]).max() < 1e-11                       # no z-velocity
    assert np.abs(S[..., :2] - S[:, :, :1, :2]).max() < 1e-10    # no z-dependence
    assert abs(S[..., 3].sum() / S[..., 3].size - 1.0) < 1e-13   # mass
    assert max(abs(S[..., d].mean()) for d in range(3)) < 1e-11  # mean momentum
    # decomposition independence: 8 boxes, same answer to solver tolerance
    S8, t8 = run(gpu, 128, nsteps, 0.0, cfl=0.7, visc_coef=visc, fixed_dt=dt)
    assert t8 == t and np.abs(S8 - S).max() < 1e-9


def test_taylor_green_bench_configuration_keeps_its_symmetry_at_256(gpu):
    """the configuration bench.py times (Tutorials/TaylorGreen/inputs.3d.taylorgreen: nu = 1e-4, cfl 0.7, prob.c = 1), two steps"""
    S, t = run(gpu, None, 2, 1.0, cfl=0.7, visc_coef=1.0e-4)
    assert t > 0 and np.abs(S[..., 0]).max() > 0.9
    # swapping x and y maps the initial data to its negative, and so does a shift by half a period in x: both images are solutions
    # with the same initial data, hence v(y, x, z) = u(x + 1/2, y, z) and w(y, x, z) = w(x + 1/2, y, z) for all times
    assert np.abs(np.roll(S[..., 0], -N // 2, axis=0) - np.swapaxes(S[..., 1], 0, 1)).max() < 1e-8
    assert np.abs(np.roll(S[..., 2], -N // 2, axis=0) - np.swapaxes(S[..., 2], 0, 1)).max() < 1e-8
    assert np.abs(S[..., 2]).max() > 1e-3                                       # the flow has become three-dimensional
    assert np.abs(S[..., 0] + S[::-1, :, :, 0]).max() < 1e-8                     # u(1 - x, y, z) = -u(x, y, z)
    assert abs(S[..., 3].sum() / S[..., 3].size - 1.0) < 1e-13
    ke = 0.5 * (S[..., :3] ** 2).sum(axis=-1).mean()
    assert 0.12 < ke < 0.125 + 1e-12                                             # kinetic energy starts at 1/8 and only decays


def test_lid_driven_cavity_at_512_on_one_gpu(gpu):
    """BASELINE config C4 at its full size on ONE MI355X (the 8 boxes of 256^3 that the 8-GPU run gives one each): the reference's
    regtest.3d.lid_driven_cavity with amr.n_cell = 512^3 and init_dt scaled with the mesh (its 0.0140625 is the 64^3 value) -- init_iter 3, the init_shrink * init_dt start from rest, two steps.
    Size-independent properties: the start-up step is init_shrink * init_dt, every solve converges, the MAC velocities are discretely
    divergence free to the solver tolerance, density stays 1 to that tolerance (conservative update with those face velocities), the lid drags
    the fluid (u > 0 next to it, bounded by the lid speed), no flow through the walls beyond O(h), slip walls at x-lo / y-lo vs no-slip at
    x-hi / y-hi break the mirror symmetry in the expected direction."""
    import os
    from iamr_amd import ns as NS
    from iamr_amd import run as R
    from iamr_amd.inputs import Inputs
    lib = gpu
    here = os.path.dirname(os.path.abspath(__file__))
    inp = Inputs([os.path.join(here, "golden", "regtest.3d.lid_driven_cavity")], ["amr.n_cell=512 512 512", "amr.max_grid_size=256", "max_step=2",
                                                                                     "ns.init_dt=0.0017578125"])      # 0.0140625 * 64 / 512
    pr = inp.problem()
    ns, lay, g, pr = R.build(inp, lib, NS, 1, pr)
    assert len(lay.boxes) == 8
    ns.post_init(pr["stop_time"])
    dts = [ns.step() for _ in range(2)]
    assert abs(dts[0] - 0.3 * 0.0017578125) < 1e-15 and dts[1] > 0
    for st in ns.stats():
        assert st.converged in (1, 2) and np.isfinite(st.resnorm)
    n = (512,) * 3
    um = [ns.data(6 + d) for d in range(3)]
    div = lib.MultiFab(lay, lib.CELL, 1, 0)
    lib.mac_divergence(g, div, um)
    umax = max(m.norm0() for m in um)
    assert umax > 0.1 and div.norm0() <= 1e-9 * umax * 512, (div.norm0(), umax)
    S = ns.data(NS.NavierStokes.S_NEW)
    lid_u, wall_w, rho_dev, vmax = [], 0.0, 0.0, 0.0
    for li in range(S.nlocal()):
        a, lo = S.to_numpy(li)
        blo, bhi, gi = lay.local_box(li)
        v = a[1:-1, 1:-1, 1:-1, :]
        assert np.isfinite(v).all()
        rho_dev = max(rho_dev, float(np.abs(v[..., 3] - 1.0).max()))
        vmax = max(vmax, float(np.abs(v[..., :3]).max()))
        if bhi[2] == 511:
            lid_u.append(v[8:-8, 8:-8, -1, 0])
            wall_w = max(wall_w, float(np.abs(v[64:-64, 64:-64, -1, 2]).max()))      # (away from the lid / side-wall corner singularities)
        if blo[2] == 0:
            wall_w = max(wall_w, float(np.abs(v[64:-64, 64:-64, 0, 2]).max()))
    assert rho_dev < 1e-9, rho_dev          # dt * (divergence left by the MAC solve at its tolerance): 1e-10 at h = 1/512, not round-off
    assert vmax <= 1.0 + 1e-9, vmax
    lid_u = np.concatenate([q.ravel() for q in lid_u])
    assert lid_u.min() > 0.0 and lid_u.mean() > 0.05, (lid_u.min(), lid_u.mean())      # the cell layer under the moving lid follows it
    assert wall_w < 0.02, wall_w                                                        # normal velocity half a cell from a wall: O(h) of the lid speed


def test_rayleigh_taylor_three_levels_on_a_256_base(gpu):
    """BASELINE config C5 at its full base size on one MI355X: the reference's regtest.3d.rayleightaylor, unmodified, with amr.n_cell = 256^3
    (variable density, gravity, Godunov_PPM, do_mom_diff, do_cons_trac, slip walls in z, max_level 2 on the vorticity, regrid every 2nd step).
    Within four coarse steps the hierarchy grows to three levels (the finest is the 1024^3 index space); composite mass of density and of the
    conservative tracer is conserved across advances, refluxes, average-downs and regrids; density keeps its two-fluid bounds; the heavy fluid
    starts to sink (mean w < 0 in the heavy layer is not assumed -- only that kinetic energy appears from rest)."""
    import os
    from iamr_amd import ns as NS
    from iamr_amd import run as R
    from iamr_amd.inputs import Inputs
    lib = gpu
    here = os.path.dirname(os.path.abspath(__file__))
    inp = Inputs([os.path.join(here, "golden", "regtest.3d.rayleightaylor")], ["amr.n_cell=256 256 256", "amr.max_grid_size=128", "max_step=4"])
    pr = inp.problem()
    amr, lays, g0 = R.build_amr(pr, lib, NS, 1)

    def composite():
        """sum over the composite grid (cells not covered by the next finer level) of density and tracer times the cell volume; min / max density; kinetic energy"""
        tot = np.zeros(2); lo_hi = [np.inf, -np.inf]; ke = 0.0
        lays_ = amr.layouts
        for l, lev in enumerate(amr.levels):
            nl = [v * 2 ** l for v in pr["n"]]
            covered = None
            if l + 1 < amr.nlev:
                # fine boxes are multiples of the blocking factor 8 in their own index space = 4 cells of this level: a lattice of 4-cell blocks
                covered = np.zeros([v // 4 for v in nl], bool)
                for blo, bhi in lays_[l + 1].boxes:
                    covered[blo[0] // 8:bhi[0] // 8 + 1, blo[1] // 8:bhi[1] // 8 + 1, blo[2] // 8:bhi[2] // 8 + 1] = True
            dx = [(pr["prob_hi"][d] - pr["prob_lo"][d]) / nl[d] for d in range(3)]
            vol = dx[0] * dx[1] * dx[2]
            S = lev.data(NS.NavierStokes.S_NEW)
            for li in range(S.nlocal()):
                a, _ = S.to_numpy(li)
                blo, bhi, gi = lays_[l].local_box(li)
                v = a[1:-1, 1:-1, 1:-1, :]
                if covered is not None:
                    m = ~covered[blo[0] // 4:bhi[0] // 4 + 1, blo[1] // 4:bhi[1] // 4 + 1, blo[2] // 4:bhi[2] // 4 + 1]
                    m = np.repeat(np.repeat(np.repeat(m, 4, 0), 4, 1), 4, 2)
                else:
                    m = np.ones(v.shape[:3], bool)
                tot += np.array([(v[..., 3] * m).sum(), (v[..., 4] * m).sum()]) * vol
                lo_hi = [min(lo_hi[0], float(v[..., 3].min())), max(lo_hi[1], float(v[..., 3].max()))]
                ke += float((0.5 * v[..., 3] * (v[..., 0] ** 2 + v[..., 1] ** 2 + v[..., 2] ** 2) * m).sum()) * vol
        return tot, lo_hi, ke

    amr.post_init(pr["stop_time"])
    m0, b0, ke0 = composite()
    levels = []
    for _ in range(4):
        amr.coarse_step()
        levels.append(amr.nlev)
    assert levels[0] == 1 and levels[-1] == 3, levels
    m4, b4, ke4 = composite()
    assert np.all(np.abs(m4 - m0) <= 1e-10 * np.abs(m0)), (m0, m4)
    assert b4[0] >= 8.44407300e+06 * (1 - 1e-3) and b4[1] <= 1.5e7 * (1 + 1e-3), b4
    assert ke4 > ke0 >= 0.0
