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

pytestmark = pytest.mark.gpu
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
