"""(a) the hydrostatic initial pressure projection of NavierStokesBase::post_init_state (NavierStokesBase.cpp:2416-2426,
Projection::initialPressureProject, Projection.cpp:841-960) and (b) the stop_time clamp of computeNewDt (NavierStokesBase.cpp:1008-1015),
HIP level driver through the C-ABI against the known answers and the CPU oracle."""
import ctypes as C
import numpy as np
import pytest

import orc

pytestmark = pytest.mark.gpu


def test_stratified_fluid_at_rest_stays_at_rest_and_matches_oracle():
    from iamr_amd import lib as L
    from iamr_amd.ns import NavierStokes, ns_params
    L.init()
    n = 16
    kw = dict(cfl=0.5, gravity=-9.8, init_iter=2, phys_lo=[0, 0, 4], phys_hi=[0, 0, 4], init_dt=0.01)
    geom = L.Geom.make([n] * 3, periodic=(1, 1, 0))
    lay = L.Layout.decompose([n] * 3, 8)
    ns = NavierStokes(geom, lay, ns_params(**kw), L.mg_opts())
    z = (np.arange(n) + 0.5) / n
    S = np.zeros((n + 2, n + 2, n + 2, 5), order="F")
    S[1:-1, 1:-1, 1:-1, 3] = (2.0 - z)[None, None, :]
    S[..., 3][S[..., 3] == 0.0] = 1.0
    mf = L.MultiFab(lay, L.CELL, 5, 1)
    mf.set_from_global(S, (-1, -1, -1))
    ns.set_data(ns.S_NEW, mf)
    ns.post_init()
    Gp = ns.data(ns.GP_NEW).gather_valid([n] * 3)
    rho = ns.data(ns.S_NEW).gather_valid([n] * 3)[..., 3]
    assert abs(Gp[..., 2] - rho * (-9.8)).max() < 1e-10 and abs(Gp[..., :2]).max() < 1e-10      # grad p = rho g
    # the oracle, same set-up
    og = orc.geom([n] * 3, periodic=(1, 1, 0))
    O = orc.lib()
    ons = C.c_void_p(O.orc_ns_create(C.byref(og), C.byref(orc.ns_params(**kw)), C.byref(orc.mg_opts())))
    So = orc.from_cfab(O.orc_ns_fab(ons, 0))
    So.valid([n] * 3)[...] = S[1:-1, 1:-1, 1:-1]
    O.orc_ns_post_init(ons, C.c_double(-1.0))
    Po = orc.from_cfab(O.orc_ns_fab(ons, 2)).valid([n] * 3, orc.NODE)[..., 0]
    P = ns.data(ns.P_NEW).gather_valid([n] * 3)[..., 0]
    # all-Neumann / periodic solve: the additive constant is not determined
    assert abs((P - P.mean()) - (Po - Po.mean())).max() <= 1e-8 * abs(Po - Po.mean()).max()
    for _ in range(2):
        ns.step()
    assert abs(ns.data(ns.S_NEW).gather_valid([n] * 3)[..., :3]).max() < 1e-9
    O.orc_ns_destroy(ons)


def test_stop_time_is_hit_exactly():
    from iamr_amd import lib as L
    from iamr_amd.ns import NavierStokes, ns_params
    L.init()
    n = 16
    ns = NavierStokes(L.Geom.make([n] * 3), L.Layout.single([n] * 3), ns_params(cfl=0.7, init_iter=1), L.mg_opts())
    ns.init_taylorgreen(c=0.0)
    stop = 0.11
    ns.post_init(stop)
    for _ in range(50):
        if ns.time >= stop - 1e-14:
            break
        ns.step()
    assert ns.time == stop
