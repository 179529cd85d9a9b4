"""GPU: physical validation against a published known answer (independent of the CPU oracle): the lid-driven cavity at Re = 100
(IAMR's regtest.3d.lid_driven_cavity set-up with nu = 0.01, lid velocity 1, unit box; periodic in y so that the flow is the classical
2-D cavity) run to a steady state; the u-velocity along the vertical centreline is compared with the table of Ghia, Ghia & Shin,
J. Comput. Phys. 48 (1982) 387-411, Table I, Re = 100.  Exercises no-slip walls, the moving lid, Godunov advection, the
Crank-Nicolson tensor solve with per-component Dirichlet BCs and the Neumann MAC / nodal projections over ~400 coupled steps."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GHIA_Y = np.array([0.0547, 0.0625, 0.0703, 0.1016, 0.1719, 0.2813, 0.4531, 0.5, 0.6172, 0.7344, 0.8516, 0.9531, 0.9609, 0.9688, 0.9766])
GHIA_U = np.array([-0.03717, -0.04192, -0.04775, -0.06434, -0.10150, -0.15662, -0.21090, -0.20581, -0.13641, 0.00332, 0.23151, 0.68717,
                   0.73722, 0.78871, 0.84123])


def test_lid_driven_cavity_re100_centreline_matches_ghia(gpu):
    lib = gpu
    from iamr_amd import ns as N
    nx = 32
    n = (nx, nx, nx)
    g = lib.Geom.make(n, periodic=(0, 1, 0))
    lid = [0.0] * 9
    lid[6] = 1.0
    ns = N.NavierStokes(g, lib.Layout.single(n), N.ns_params(cfl=0.8, visc_coef=0.01, init_dt=0.3 / nx, init_shrink=0.3, init_iter=3,
                                                              phys_lo=[5, 0, 5], phys_hi=[5, 0, 5], wall_vel_hi=lid))
    ns.init_rest(1.0)
    ns.post_init(-1.0)
    while ns.time < 10.0:
        ns.step()
    u = ns.data(N.NavierStokes.S_NEW).gather_valid(n)[..., 0]
    assert np.abs(u - u[:, :1, :]).max() < 1e-10            # the 3-D run stays two-dimensional
    z = (np.arange(nx) + 0.5) / nx
    uc = 0.5 * (u[nx // 2 - 1, 0, :] + u[nx // 2, 0, :])    # x = 0.5
    ui = np.interp(GHIA_Y, z, uc)
    # 32^2 cells, second-order scheme, t = 10 (steady to ~1e-3): measured max deviation 0.005; Ghia's own table has 5 digits
    assert np.abs(ui - GHIA_U).max() < 0.012, np.abs(ui - GHIA_U).max()
    assert abs(uc.min() - (-0.2109)) < 0.006
