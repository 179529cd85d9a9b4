"""Pins of the BDS restatement (oracle/orc_bds.c; ns.advection_scheme = BDS, Source/NavierStokesBase.cpp:548-553, 4701-4717) that need
neither a GPU nor AMReX-Hydro (absent from the reference tree): known answers of the published algorithm (Nonaka, May, Almgren, Bell, SISC
33, 2011; Docs/sphinx_documentation/source/TimeStep.rst:92-133).

* a trilinear profile under a constant velocity field is advected EXACTLY: the edge state is the profile at the centre of the swept
  region, x + h/2 - u dt/2 (the limited slopes reproduce a trilinear function, the space-time integrals are exact for it);
* the update is conservative; a uniform state stays uniform in a divergence-free variable velocity field;
* second-order convergence of one advance of a smooth profile; boundedness (no new extrema to solver round-off) for a step profile
  over many steps with the slope limiter, where the unlimited fourth-order interpolant overshoots."""
import ctypes as C
import numpy as np
import pytest

import orc

BDS = 2


@pytest.fixture(autouse=True)
def _scheme():
    orc.lib().orc_godunov_set_ppm(0)
    yield
    orc.lib().orc_godunov_set_ppm(0)


def _periodic(a, ng):
    return np.pad(a, [(ng, ng)] * 3 + [(0, 0)] * (a.ndim - 3), mode="wrap")


def _aofs(n, S, mac, dt, iconserv, dx=None):
    """-> (aofs, edge states) of orc_compute_aofs with BDS on a periodic box; S: (n.., ncomp) valid cells, mac: three face arrays (valid
    faces incl. the periodic duplicate)"""
    L = orc.lib()
    L.orc_godunov_set_ppm(BDS)
    g = orc.geom(n)
    nc = S.shape[-1]
    Sf = orc.Fab(n, orc.CELL, 3, nc)
    Sf.a[...] = _periodic(S, 3)
    um = []
    for d in range(3):
        f = orc.Fab(n, orc.face(d), 1, 1)
        core = mac[d]
        sl = [slice(None)] * 3
        sl[d] = slice(0, n[d])
        f.a[..., 0] = np.pad(core[tuple(sl)], [(1, 2 if e == d else 1) for e in range(3)], mode="wrap")
        um.append(f)
    aofs = orc.Fab(n, orc.CELL, 0, nc)
    edge = [orc.Fab(n, orc.face(d), 0, nc) for d in range(3)]
    ic = (C.c_int * nc)(*iconserv)
    L.orc_compute_aofs(C.byref(g), aofs.ref(), 0, Sf.ref(), nc, None, None, orc.fabptrs(um), ic, C.c_double(dt), orc.bcrecs(nc), 0, 0, orc.fabptrs(edge), None)
    return aofs.a.copy(), [e.a.copy() for e in edge]


def _const_mac(n, vel):
    return [np.full(tuple(n[e] + (1 if e == d else 0) for e in range(3)), vel[d]) for d in range(3)]


def test_trilinear_profile_is_advected_exactly():
    n = (8, 8, 8)
    h = 1.0 / 8
    vel = (0.7, -0.4, 0.25)
    dt = 0.6 * h
    c = [(np.arange(n[d]) + 0.5) * h for d in range(3)]
    X, Y, Z = np.meshgrid(*c, indexing="ij")
    f = lambda x, y, z: 1.0 + 2.0 * x - 1.5 * y + 0.5 * z + 0.8 * x * y - 0.6 * x * z + 0.4 * y * z + 1.2 * x * y * z
    S = f(X, Y, Z)[..., None]
    aofs, edge = _aofs(n, S, _const_mac(n, vel), dt, (1,))
    # faces away from the periodic seam (the profile is not periodic; cells 2..5 and their slopes do not see it): faces 3, 4
    for d in range(3):
        sl = tuple(slice(3, 5) for _ in range(3))
        xf = [c[e][3:5] for e in range(3)]
        xf[d] = (np.arange(3, 5)) * h                                    # face positions
        Xf, Yf, Zf = np.meshgrid(*xf, indexing="ij")
        # domain of dependence of the face over dt: the centroid is the face centre shifted by -vel dt / 2 (exact for trilinear data up to
        # the second moments of the swept parallelepiped, which the algorithm's quadratures integrate exactly)
        p = [Xf - 0.5 * vel[0] * dt, Yf - 0.5 * vel[1] * dt, Zf - 0.5 * vel[2] * dt]
        exact = f(*p)
        # second moments of the traced region: the average of x y over a parallelepiped swept along (u, v, w) dt adds u v dt^2 / 12 etc.
        exact += (0.8 * vel[0] * vel[1] - 0.6 * vel[0] * vel[2] + 0.4 * vel[1] * vel[2]) * dt * dt / 12.0
        exact += 1.2 * dt * dt / 12.0 * (vel[0] * vel[1] * p[2] + vel[0] * vel[2] * p[1] + vel[1] * vel[2] * p[0])
        got = edge[d][sl + (0,)]
        assert np.abs(got - exact).max() < 1e-12, (d, np.abs(got - exact).max())


def test_conservation_and_free_stream():
    rng = np.random.default_rng(5)
    n = (8, 8, 8)
    h = 1.0 / 8
    # discretely divergence-free face velocities from a vector potential on edges
    A = [rng.standard_normal((9, 9, 9)) for _ in range(3)]
    for a in A:
        for d in range(3):
            sl0 = [slice(None)] * 3; sl1 = [slice(None)] * 3
            sl0[d], sl1[d] = 0, 8
            a[tuple(sl1)] = a[tuple(sl0)]
    u = (A[2][:, 1:, :-1] - A[2][:, :-1, :-1]) / h - (A[1][:, :-1, 1:] - A[1][:, :-1, :-1]) / h
    v = (A[0][:-1, :, 1:] - A[0][:-1, :, :-1]) / h - (A[2][1:, :, :-1] - A[2][:-1, :, :-1]) / h
    w = (A[1][1:, :-1, :] - A[1][:-1, :-1, :]) / h - (A[0][:-1, 1:, :] - A[0][:-1, :-1, :]) / h
    mac = [0.05 * u, 0.05 * v, 0.05 * w]
    div = (mac[0][1:] - mac[0][:-1] + mac[1][:, 1:] - mac[1][:, :-1] + mac[2][:, :, 1:] - mac[2][:, :, :-1]) / h
    assert np.abs(div).max() < 1e-12
    dt = 0.5 * h / max(np.abs(m).max() for m in mac)
    S = np.stack([np.ones(n), rng.uniform(1.0, 2.0, n)], axis=-1)
    aofs, _ = _aofs(n, S, mac, dt, (0, 1))
    assert np.abs(aofs[..., 0]).max() < 1e-12                           # uniform state, convective form: stays uniform
    assert abs(aofs[..., 1].sum()) < 1e-11 * np.abs(aofs[..., 1]).sum()  # conservative form: fluxes telescope
    aofs_c, _ = _aofs(n, S, mac, dt, (1, 1))
    assert np.abs(aofs_c[..., 0]).max() < 1e-12                         # uniform state, conservative form in a solenoidal field


def test_second_order_convergence():
    errs = []
    for m in (16, 32):
        n = (m, m, m)
        h = 1.0 / m
        vel = (1.0, 0.5, -0.75)
        dt = 0.4 * h
        c = [(np.arange(m) + 0.5) * h for _ in range(3)]
        X, Y, Z = np.meshgrid(*c, indexing="ij")
        f = lambda x, y, z: np.sin(2 * np.pi * x) * np.cos(2 * np.pi * y) + 0.5 * np.sin(2 * np.pi * (y + z))
        # cell averages of f (exact to O(h^4) by the fourth-order correction of the point value)
        lap = -(2 * np.pi) ** 2 * (2 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) + 0.5 * 2 * np.sin(2 * np.pi * (Y + Z)))
        S = (f(X, Y, Z) + h * h / 24.0 * lap)[..., None]
        aofs, _ = _aofs(n, S, _const_mac(n, vel), dt, (1,))
        new = S[..., 0] - dt * aofs[..., 0]
        p = (X - vel[0] * dt, Y - vel[1] * dt, Z - vel[2] * dt)
        lap_p = -(2 * np.pi) ** 2 * (2 * np.sin(2 * np.pi * p[0]) * np.cos(2 * np.pi * p[1]) + 0.5 * 2 * np.sin(2 * np.pi * (p[1] + p[2])))
        exact = f(*p) + h * h / 24.0 * lap_p
        errs.append(np.abs(new - exact).mean() / dt)                    # local truncation error, L1 (the limiter clips smooth extrema: the maximum norm sits there)
    assert errs[1] < errs[0] / 3.3, errs                                # second order in the truncation error


def test_step_profile_stays_bounded():
    n = (16, 16, 4)
    h = 1.0 / 16
    vel = (1.0, 0.6, 0.0)
    dt = 0.8 * h / 1.0
    S = np.zeros(n + (1,))
    S[4:10, 5:11, :, 0] = 1.0
    mac = _const_mac(n, vel)
    # h differs in z (n = 4 cells of a unit box): irrelevant with w = 0 and z-uniform data
    for step in range(12):
        aofs, _ = _aofs(n, S, mac, dt, (1,))
        S = S - dt * aofs
        assert S.min() > -1e-9 and S.max() < 1.0 + 1e-9, (step, S.min(), S.max())     # the limiter's own threshold is 1e-10
    assert abs(S.sum() - 36.0 * 4) < 1e-10
