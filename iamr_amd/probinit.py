"""Initial data of IAMR's closed-form problem set-ups that have no device initialiser in libiamrx (host side, numpy; the state is
handed to the library once through iamrx_ns_set_data).  Restates Source/prob/prob_init.cpp for
  probtype 4  constant velocity + tanh tracer blob        (init_constant_vel_rho, :232-281; the Poiseuille / channel regtests)
  probtype 5  DoubleShearLayer, uniform in z              (init_DoubleShearLayer, :346-405)
  probtype 7  Euler: vortex tube along x with a wobble    (init_Euler, :562-610)
Cell centres x = prob_lo + (i + 1/2) dx as in the reference; components (u, v, w, rho, tracer)."""
import numpy as np


def cell_centres(n, prob_lo, prob_hi, lo=(0, 0, 0), hi=None):
    hi = tuple(v - 1 for v in n) if hi is None else hi
    c = []
    for d in range(3):
        dx = (prob_hi[d] - prob_lo[d]) / n[d]
        c.append(prob_lo[d] + (np.arange(lo[d], hi[d] + 1) + 0.5) * dx)
    return np.meshgrid(*c, indexing="ij")


def initial_state(prob, X, Y, Z, nstate=5):
    """prob: the dict Inputs.problem()['prob'] builds; X, Y, Z: cell-centre coordinates (3-D arrays) -> array (..., nstate);
    components past the tracer (tracer2, temp) as prob_init.cpp's `for nt = 2 .. nscal-1` loops set them"""
    S = np.zeros(X.shape + (nstate,))
    pt = prob["probtype"]
    if prob.get("dim", 3) == 2:
        # a two-dimensional problem on the (x, z) plane of a y-periodic slab (inputs.Inputs.lift_2d): the AMREX_SPACEDIM == 2 branches of
        # prob_init.cpp see (x, y) = our (X, Z), their v is our w, distances have no third term
        S2 = initial_state(dict(prob, dim=3, _plane=True), X, Z, np.zeros_like(Z), nstate)
        S[..., 0], S[..., 2] = S2[..., 0], S2[..., 1]
        S[..., 3:] = S2[..., 3:]
        return S
    if prob.get("_plane"):                       # evaluated by the 2-D branch above: Z is zero, the blob centre's third entry is ignored
        prob = dict(prob, blob_center=[prob["blob_center"][0], prob["blob_center"][2], 0.0], velocity_ic=[prob["velocity_ic"][0], prob["velocity_ic"][2], 0.0])
    extra = 1.0                              # prob_init.cpp:400-403 (5), 605-608 (7)
    if pt in (2, 6):                         # init_bubble, prob_init.cpp:164-229; 6: hot bubble with temperature as the last scalar
        v = prob["velocity_ic"]
        S[..., 0], S[..., 1], S[..., 2] = v[0], v[1], v[2]
        bc = prob["blob_center"]
        dist = np.sqrt((X - bc[0]) ** 2 + (Y - bc[1]) ** 2 + (Z - bc[2]) ** 2)
        blob = np.where(dist < prob["blob_radius"], 1.0, 0.0)
        S[..., 4:] = blob[..., None]
        if pt == 6:
            S[..., 3] = 1.0 / prob["density_ic"] + 0.5 * (1.0 - 1.0 / prob["density_ic"]) * (1.0 + np.tanh(40.0 * (dist - prob["blob_radius"]) / prob["interface_width"]))
            S[..., nstate - 1] = 1.0 / S[..., 3]
        else:
            S[..., 3] = 1.0 + 0.5 * (prob["density_ic"] - 1.0) * (1.0 - np.tanh(30.0 * (dist - prob["blob_radius"]) / prob["interface_width"]))
        return S
    if pt == 4:
        extra = 0.0                          # prob_init.cpp:275-279
        v = prob["velocity_ic"]
        S[..., 0], S[..., 1], S[..., 2] = v[0], v[1], v[2]
        bc = prob["blob_center"]
        dist = np.sqrt((X - bc[0]) ** 2 + (Y - bc[1]) ** 2 + (Z - bc[2]) ** 2)
        S[..., 3] = prob["density_ic"]
        S[..., 4] = 0.5 * (1.0 - np.tanh(25.0 * (dist - prob["blob_radius"]) / prob["interface_width"]))
    elif pt == 5:
        w = prob["interface_width"]
        if prob["direction"] == 1:          # shear layer in y
            S[..., 0] = -0.05 * np.sin(np.pi * Y)
            S[..., 1] = np.tanh(30.0 * (0.5 - np.abs(X)) / w)
        elif prob["direction"] == 0:        # shear layer in x
            S[..., 0] = np.tanh(30.0 * (0.5 - np.abs(Y)) / w)
            S[..., 1] = 0.05 * np.sin(np.pi * X)
        else:
            raise ValueError("DoubleShearLayer: prob.direction must be 0 or 1 (prob_init.cpp:359-360)")
        bc = prob["blob_center"]
        dist = np.sqrt((X - bc[0]) ** 2 + (Y - bc[1]) ** 2 + (Z - bc[2]) ** 2)
        S[..., 3] = prob["density_ic"]
        S[..., 4] = np.where(dist < prob["blob_radius"], 1.0, 0.0)
    elif pt == 7:
        eps, rho_in, beta, delta, kappa = 0.05, 0.15, 15.0, 0.0333, 500.0
        x, y, z = X - 0.5, Y - 0.5, Z - 0.5
        r = np.sqrt(y * y + z * z)
        S[..., 0] = np.tanh((rho_in - r) / delta)
        S[..., 2] = eps * np.exp(-beta * (x * x + y * y))
        S[..., 3] = prob["density_ic"]
        S[..., 4] = np.exp(-kappa * (rho_in - r) ** 2)
    else:
        raise NotImplementedError(f"probinit: prob.probtype = {pt}")
    S[..., 5:] = extra
    return S


def set_initial_state(ns, lay, lib, N, prob, n, prob_lo, prob_hi):
    """fill S_new of `ns` (valid cells; post_init fills the ghost cells) on every local box"""
    nstate = ns.nstate
    m = lib.MultiFab(lay, lib.CELL, nstate, 1)
    for li in range(m.nlocal()):
        lo, hi = m.fab_box(li)
        X, Y, Z = cell_centres(n, prob_lo, prob_hi, lo, hi)
        m.from_numpy(initial_state(prob, X, Y, Z, nstate), li)
    ns.set_data(N.NavierStokes.S_NEW, m)
