"""GPU: the register-resident plane-fused nodal Gauss-Seidel pass (k_nodal_gsr: 64 x 64 footprints, planes in registers, x-neighbours
by DPP lane shifts, y-neighbours through LDS rows) produces the doubles of the 8 sequential colour passes (k_nodal_gscolor) and of the
LDS-staged k_nodal_gs4 -- in every variant: index wrap on a periodic box, ghost fills between boxes and at walls, Dirichlet masks
(outflow faces, a level that does not cover the domain), constant sigma; boxes with odd lower corners, tiles narrower than the
footprint, several z-chunks per tile, both patch heights (GSR_PB 4 / 8)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
PERIODIC, DIRICHLET, NEUMANN = 0, 101, 102


class tuning:
    def __init__(self, lib, **kv):
        self.lib, self.kv = lib, kv

    def __enter__(self):
        self.old = {k: self.lib.tuning_get(k, d) for k, (v, d) in self.kv.items()}
        for k, (v, d) in self.kv.items():
            self.lib.tuning_set(k, v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            self.lib.tuning_set(k, v)


def periodic_fields(n, seed):
    rng = np.random.default_rng(seed)
    Xg = rng.standard_normal(n)
    Rg = rng.standard_normal(n)
    Sg = 1.0 / (1.0 + 0.8 * rng.random(n))

    def node_field(G, ng):
        idx = [np.mod(np.arange(-ng, n[d] + 1 + ng), n[d]) for d in range(3)]
        return G[np.ix_(*idx)][..., None]

    def cell_field(G, ng):
        idx = [np.mod(np.arange(-ng, n[d] + ng), n[d]) for d in range(3)]
        return G[np.ix_(*idx)][..., None]
    return Xg, Rg, Sg, node_field, cell_field


# (160, 128, 96): more workgroups than the chip holds at once in the old kernel; (112, 60, 20): partial tiles in both directions;
# (128, 128, 64) in 64^3 boxes: ghost fills between boxes, two tiles per box; (198, 150, 18) in boxes of (99, 75, 9): odd lower corners
@pytest.mark.parametrize("pb", [4, 8])
@pytest.mark.parametrize("n,boxes", [((64, 64, 16), None), ((112, 60, 20), None), ((160, 128, 96), None), ((128, 128, 64), (64, 64, 32)),
                                     ((198, 150, 18), (99, 75, 9)), ((264, 256, 12), None)])
def test_sweep_equals_eight_colour_passes(gpu, n, boxes, pb):
    lib = gpu
    from iamr_amd import ns as N
    g = lib.Geom.make(n)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    Xg, Rg, Sg, node_field, cell_field = periodic_fields(n, 11)
    res = {}
    for tag, fused, gsr in (("colour", 0, 0), ("gs4", 1, 0), ("gsr", 1, 1)):
        with tuning(lib, GSR=(gsr, 1), GSR_PB=(pb, 4), GSR_MIN=(48, 48)):
            ng = 4
            x = lib.MultiFab(lay, lib.NODE, 1, ng)
            r = lib.MultiFab(lay, lib.NODE, 1, ng)
            s = lib.MultiFab(lay, lib.CELL, 1, ng)
            x.set_from_global(node_field(Xg, ng), (-ng,) * 3)
            r.set_from_global(node_field(Rg, ng), (-ng,) * 3)
            s.set_from_global(cell_field(Sg, ng), (-ng,) * 3)
            for _ in range(2):
                N.nodal_gs_sweep(g, x, r, s, fused)
            x.fill_boundary(g)
            res[tag] = x.gather_valid(n)
    assert np.isfinite(res["gsr"]).all()
    assert np.array_equal(res["colour"], res["gs4"]), np.abs(res["colour"] - res["gs4"]).max()
    assert np.array_equal(res["colour"], res["gsr"]), np.abs(res["colour"] - res["gsr"]).max()


def solve_case(lib, n, per, lobc, hibc, boxes, sigma_const, seed, cover=None, iters=2):
    """phi after `iters` multigrid iterations of iamrx_nodal_solve (upstream cycle shape: 4 sweeps x (2 + 2) smooth calls per level)"""
    from iamr_amd import ns as N
    import orc
    rng = np.random.default_rng(seed)
    g = lib.Geom.make(n, periodic=per)
    lay = lib.Layout(boxes)
    shape_n = tuple(n[d] + 3 for d in range(3)) + (1,)
    shape_c = tuple(n[d] + 2 for d in range(3)) + (1,)
    sig = np.full(shape_c, 0.75) if sigma_const else 0.5 + rng.random(shape_c)
    rhs = rng.standard_normal(tuple(n[d] + 1 for d in range(3)) + (1,))
    phi = rng.standard_normal(shape_n)
    for d in range(3):
        if per[d]:
            for a, off in ((rhs, 0), (phi, 1)):
                hi = [slice(None)] * 4; lo = [slice(None)] * 4
                hi[d] = off + n[d]; lo[d] = off
                a[tuple(hi)] = a[tuple(lo)]
    sig_d = lib.MultiFab(lay, lib.CELL, 1, 1); sig_d.set_from_global(sig, (-1,) * 3)
    rhs_d = lib.MultiFab(lay, lib.NODE, 1, 0); rhs_d.set_from_global(rhs, (0,) * 3)
    phi_d = lib.MultiFab(lay, lib.NODE, 1, 1); phi_d.set_from_global(phi, (-1,) * 3)
    N.nodal_solve(g, phi_d, rhs_d, sig_d, 0, lobc, hibc, 1e-30, 0.0, lib.mg_opts(fixed_iters=iters, **orc.UPSTREAM_NODAL_CYCLE))
    return [phi_d.to_numpy(li)[0].copy() for li in range(phi_d.nlocal())]


CASES = {
    # one box spanning a periodic domain: index wrap, no ghost fills
    "wrap": dict(n=(96, 64, 32), per=(1, 1, 1), lobc=(PERIODIC,) * 3, hibc=(PERIODIC,) * 3, boxes=[((0, 0, 0), (95, 63, 31))], sigma_const=False),
    "wrap_csig": dict(n=(96, 64, 32), per=(1, 1, 1), lobc=(PERIODIC,) * 3, hibc=(PERIODIC,) * 3, boxes=[((0, 0, 0), (95, 63, 31))], sigma_const=True),
    # Neumann walls: ghost nodes by reflection, recomputed in the halo
    "walls": dict(n=(64, 96, 16), per=(0, 0, 0), lobc=(NEUMANN,) * 3, hibc=(NEUMANN,) * 3, boxes=[((0, 0, 0), (63, 95, 15))], sigma_const=False),
    "walls_csig_2box": dict(n=(128, 64, 16), per=(0, 1, 0), lobc=(NEUMANN, PERIODIC, NEUMANN), hibc=(NEUMANN, PERIODIC, NEUMANN),
                            boxes=[((0, 0, 0), (63, 63, 15)), ((64, 0, 0), (127, 63, 15))], sigma_const=True),
    # outflow face: Dirichlet mask
    "outflow": dict(n=(64, 64, 16), per=(0, 1, 0), lobc=(NEUMANN, PERIODIC, NEUMANN), hibc=(DIRICHLET, PERIODIC, NEUMANN),
                    boxes=[((0, 0, 0), (63, 63, 15))], sigma_const=False),
    # a level that covers part of the domain (refined AMR level): its boundary nodes inside the domain are Dirichlet nodes
    "patch": dict(n=(128, 128, 32), per=(1, 1, 1), lobc=(PERIODIC,) * 3, hibc=(PERIODIC,) * 3,
                  boxes=[((24, 32, 8), (87, 95, 23)), ((88, 32, 8), (111, 95, 23))], sigma_const=False),
}


@pytest.mark.parametrize("pb", [4, 8])
@pytest.mark.parametrize("case", sorted(CASES))
def test_solver_iterates_identical_to_lds_kernel(gpu, case, pb):
    lib = gpu
    kw = CASES[case]
    with tuning(lib, GSR=(0, 1)):
        ref = solve_case(lib, seed=3, **kw)
    with tuning(lib, GSR=(1, 1), GSR_PB=(pb, 4), GSR_MIN=(48, 48)):
        got = solve_case(lib, seed=3, **kw)
    for a, b in zip(ref, got):
        assert np.isfinite(b).all()
        assert np.array_equal(a, b), (case, float(np.abs(a - b).max()))


@pytest.mark.parametrize("n,per", [((128, 96, 64), (0, 0, 0)), ((96, 64, 128), (0, 1, 0)), ((64, 64, 64), (1, 0, 0)), ((32, 24, 16), (0, 0, 0))])
def test_mirror_images_at_neumann_walls_equal_the_ghost_fills(gpu, n, per):
    """round 5: one box spanning a domain whose non-periodic directions end on Neumann walls (the pressure of LidDrivenCavity, of a channel):
    the index-wrap variants of k_nodal_gsr / k_nodal_gs4 take the MIRROR image of a node / cell beyond a wall (image_node / image_cell) --
    the value nodal_reflect_bc / cc_mirror_bc would have written into the ghost node / cell -- and no ghost fill runs in front of a pass
    (Projection.cpp:2385-2567's solve on a closed domain; mlndlap_fillbc_cc / mlndlap_applybc roles).  The same doubles: a whole nodal
    projection under IAMRX_NODAL_REFLECT_WRAP = 1 / 0 gives the same iterations, pressure and velocity bit for bit (the ghost-fill form
    is the one the oracle tests of tests/test_gpu_walls.py pin)."""
    lib = gpu
    from iamr_amd import ns as N
    g = lib.Geom.make(n, prob_hi=tuple(v / n[0] for v in n), periodic=per)
    lay = lib.Layout.single(n)
    lobc = tuple(PERIODIC if per[d] else NEUMANN for d in range(3))
    rng = np.random.default_rng(8)
    ax = [(np.arange(-1, n[d] + 1) + 0.5) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    rho = 1.0 + 0.4 * np.sin(2 * np.pi * X) * np.cos(2 * np.pi * Y) * np.cos(2 * np.pi * Z) + 0.1 * rng.random(X.shape)
    vel = np.zeros(X.shape + (3,))
    vel[..., 0] = np.sin(np.pi * X) * np.cos(2 * np.pi * Y) + 0.1 * rng.standard_normal(X.shape)
    vel[..., 1] = np.sin(np.pi * Y) * np.cos(2 * np.pi * Z) + 0.1 * rng.standard_normal(X.shape)
    vel[..., 2] = np.sin(np.pi * Z) * np.cos(2 * np.pi * X) + 0.1 * rng.standard_normal(X.shape)
    for d in range(3):
        sl0 = [slice(None)] * 3; sl1 = [slice(None)] * 3; sh0 = [slice(None)] * 3; sh1 = [slice(None)] * 3
        sl0[d] = 0; sl1[d] = 1; sh0[d] = -1; sh1[d] = -2
        if per[d]:
            s0 = [slice(None)] * 3; s1 = [slice(None)] * 3
            s0[d] = n[d]; s1[d] = 1
            vel[tuple(sl0)] = vel[tuple(s0)]; vel[tuple(sh0)] = vel[tuple(s1)]
            rho[tuple(sl0)] = rho[tuple(s0)]; rho[tuple(sh0)] = rho[tuple(s1)]
        else:
            vel[tuple(sl0)] = vel[tuple(sl1)]; vel[tuple(sh0)] = vel[tuple(sh1)]
            vel[tuple(sl0) + (d,)] = -vel[tuple(sl1) + (d,)]; vel[tuple(sh0) + (d,)] = -vel[tuple(sh1) + (d,)]      # no flow through the walls
            rho[tuple(sl0)] = rho[tuple(sl1)]; rho[tuple(sh0)] = rho[tuple(sh1)]
    out = {}
    for mode in (1, 0):
        with tuning(lib, NODAL_REFLECT_WRAP=(mode, 1)):
            sig_d = lib.MultiFab(lay, lib.CELL, 1, 1); sig_d.set_from_global((1.0 / rho)[..., None], (-1,) * 3)
            vel_d = lib.MultiFab(lay, lib.CELL, 3, 1); vel_d.set_from_global(vel, (-1,) * 3)
            p_d = lib.MultiFab(lay, lib.NODE, 1, 1); p_d.setval(0.0)
            st = N.nodal_projection(g, vel_d, 0, p_d, sig_d, 0, lobc=lobc, hibc=lobc, rel_tol=1e-10, abs_tol=1e-16)
            assert st.converged >= 1
            out[mode] = (st.iters, st.resnorm, vel_d.gather_valid(n), p_d.gather_valid(n))
    assert out[1][0] == out[0][0] and out[1][1] == out[0][1], (out[1][:2], out[0][:2])
    assert np.array_equal(out[1][3], out[0][3]), float(np.abs(out[1][3] - out[0][3]).max())
    assert np.array_equal(out[1][2], out[0][2])


@pytest.mark.boxes_kept
@pytest.mark.parametrize("case", ["periodic", "walls", "patch"])
def test_pass_issued_in_two_parts_on_two_streams_gives_the_same_doubles(gpu, case):
    """round 6: IAMRX_HALO_OVERLAP = 2 forces what a multi-rank run does by itself -- every pass of k_nodal_gsr whose
    ghost nodes are refreshed first is issued as the tiles that read no ghost node on the main stream, and the exchange + wall reflection +
    the remaining tiles on the side stream behind a fork, joined afterwards.  Same doubles as the one-piece pass (HALO_OVERLAP = 0), on two
    boxes of 128 x 128 cells in-plane (3 x 3 tiles each: one interior tile per box and z-chunk), with walls, and on a refined patch (mask)."""
    lib = gpu
    if case == "periodic":
        kw = dict(n=(256, 128, 64), per=(1, 1, 1), lobc=(PERIODIC,) * 3, hibc=(PERIODIC,) * 3, boxes=[((0, 0, 0), (127, 127, 63)), ((128, 0, 0), (255, 127, 63))])
    elif case == "walls":
        kw = dict(n=(128, 256, 48), per=(0, 0, 0), lobc=(NEUMANN,) * 3, hibc=(NEUMANN,) * 3, boxes=[((0, 0, 0), (127, 127, 47)), ((0, 128, 0), (127, 255, 47))])
    else:
        kw = dict(n=(320, 192, 64), per=(1, 1, 1), lobc=(PERIODIC,) * 3, hibc=(PERIODIC,) * 3, boxes=[((32, 32, 8), (159, 159, 55)), ((160, 32, 8), (287, 159, 55))])
    out = {}
    for ov in (0, 2):
        lib.tuning_set("HALO_OVERLAP", ov)
        try:
            out[ov] = solve_case(lib, seed=5, sigma_const=False, iters=2, **kw)
        finally:
            lib.tuning_set("HALO_OVERLAP", 1)
    for a, b in zip(out[0], out[2]):
        assert np.isfinite(a).all() and np.array_equal(a, b), float(np.abs(a - b).max())
