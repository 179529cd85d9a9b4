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
