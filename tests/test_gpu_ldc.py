"""NavierStokes::advance + init sequence with physical walls (BASELINE config C4, LidDrivenCavity, reduced to sizes the
CPU oracle finishes in seconds): slip / no-slip walls, moving lid, tensor diffusion with per-component BCs, Neumann MAC and
nodal projections, tracer diffusion (Diffusion::diffuse_scalar), init_dt start from rest.
Reference inputs: Exec/run3d/regtest.3d.lid_driven_cavity:5-46."""
import ctypes as C

import os
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

LDC = dict(cfl=0.3, visc_coef=0.01, init_dt=0.0140625, init_shrink=0.3, init_iter=3, tracer_diff_coef=0.001)


def oracle_params(orc, per, phys_lo, phys_hi, lid, **kw):
    L = orc.lib()
    p = orc.CNsParams()
    L.orc_ns_default_params(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    for d in range(3):
        p.phys_lo[d] = 0 if per[d] else phys_lo[d]
        p.phys_hi[d] = 0 if per[d] else phys_hi[d]
    for q in range(9):
        p.wall_vel_lo[q] = 0.0
        p.wall_vel_hi[q] = lid[q]
    return p


def run_oracle(orc, n, per, phys_lo, phys_hi, lid, nsteps, init, prob_hi=(1.0, 1.0, 1.0), **kw):
    L = orc.lib()
    g = orc.geom(n, probhi=prob_hi, periodic=per)
    p = oracle_params(orc, per, phys_lo, phys_hi, lid, **kw)
    o = orc.mg_opts()
    s = C.c_void_p(L.orc_ns_create(C.byref(g), C.byref(p), C.byref(o)))
    assert s.value
    L.orc_ns_init_rest(s, C.c_double(1.0))
    if init is not None:
        Sn = orc.from_cfab(L.orc_ns_fab(s, 0))
        Sn.a[1:-1, 1:-1, 1:-1, :] = init
    L.orc_ns_post_init(s, C.c_double(-1.0))
    dts = [L.orc_ns_step(s) for _ in range(nsteps)]
    S = orc.from_cfab(L.orc_ns_fab(s, 0)).valid(n).copy()
    Pn = orc.from_cfab(L.orc_ns_fab(s, 2)).valid(n, orc.NODE).copy()
    Gp = orc.from_cfab(L.orc_ns_fab(s, 4)).a.copy()
    st = [orc.CMgStats() for _ in range(3)]
    L.orc_ns_last_stats(s, C.byref(st[0]), C.byref(st[1]), C.byref(st[2]))
    T = L.orc_ns_time(s)
    L.orc_ns_destroy(s)
    return S, Pn, Gp, T, dts, st


def run_gpu(lib, n, per, phys_lo, phys_hi, lid, nsteps, init, boxes, prob_hi=(1.0, 1.0, 1.0), **kw):
    from iamr_amd import ns as N
    g = lib.Geom.make(n, prob_hi=prob_hi, periodic=per)
    lay = lib.Layout.decompose(n, boxes) if boxes else lib.Layout.single(n)
    plo = [0 if per[d] else phys_lo[d] for d in range(3)]
    phi = [0 if per[d] else phys_hi[d] for d in range(3)]
    ns = N.NavierStokes(g, lay, N.ns_params(phys_lo=plo, phys_hi=phi, wall_vel_hi=lid, **kw))
    ns.init_rest(1.0)
    if init is not None:
        m = lib.MultiFab(lay, lib.CELL, 5, 1)
        G = np.zeros(tuple(v + 2 for v in n) + (5,))
        G[1:-1, 1:-1, 1:-1, :] = init
        m.set_from_global(G, (-1, -1, -1))
        ns.set_data(N.NavierStokes.S_NEW, m)
    ns.post_init(-1.0)
    dts = [ns.step() for _ in range(nsteps)]
    return ns, lay, g, dts


def compare(lib, ns, lay, g, n, dts, ref, tol=1e-8):
    from iamr_amd import ns as N
    S_o, P_o, Gp_o, T_o, dts_o, st_o = ref
    assert np.allclose(dts, dts_o, rtol=1e-9, atol=0.0)
    assert abs(ns.time - T_o) <= 1e-12
    S = ns.data(N.NavierStokes.S_NEW).gather_valid(n)
    for comp in range(5):
        scale = max(np.abs(S_o[..., comp]).max(), 1e-3)
        assert np.abs(S[..., comp] - S_o[..., comp]).max() <= tol * scale, comp
    Pd = ns.data(N.NavierStokes.P_NEW).gather_valid(n)[..., 0]
    Pr = P_o[..., 0]
    assert np.abs((Pd - Pd.mean()) - (Pr - Pr.mean())).max() <= 1e-6 * max(np.abs(Pr - Pr.mean()).max(), 1e-3)
    um = [ns.data(6 + d) for d in range(3)]
    div = lib.MultiFab(lay, lib.CELL, 1, 0)
    lib.mac_divergence(g, div, um)
    assert div.norm0() <= 1e-9
    return S


LID = [0.0] * 9
LID[2 * 3 + 0] = 1.0       # zhi.velocity = 1 0 0


@pytest.mark.parametrize("boxes", [None, 8])
def test_lid_driven_cavity_matches_oracle(orc, gpu, boxes):
    """regtest.3d.lid_driven_cavity at 16^3: start from rest (init_dt), lo_bc = 4 4 5, hi_bc = 5 5 5, lid u = 1,
    3 pressure iterations + 4 steps.  Tolerance 1e-8 relative (solver tolerances 1e-12 / 1e-10, reduction order)."""
    n = (16, 16, 16)
    per = (0, 0, 0)
    ref = run_oracle(orc, n, per, (4, 4, 5), (5, 5, 5), LID, 4, None, **LDC)
    ns, lay, g, dts = run_gpu(gpu, n, per, (4, 4, 5), (5, 5, 5), LID, 4, None, boxes, **LDC)
    S = compare(gpu, ns, lay, g, n, dts, ref)
    assert abs(dts[0] - 0.3 * 0.0140625) < 1e-15          # init_shrink * init_dt
    assert np.abs(S[..., 0]).max() > 0.05                 # the lid drags the fluid
    # Gp ghost cells: foextrap copy of the first interior cell at every wall
    from iamr_amd import ns as N
    if boxes is None:
        gp, lo = ns.data(N.NavierStokes.GP_NEW).to_numpy(0)
        assert np.array_equal(gp[0, 1:-1, 1:-1], gp[1, 1:-1, 1:-1]) and np.array_equal(gp[1:-1, 1:-1, -1], gp[1:-1, 1:-1, -2])


def test_walls_general_state_and_tracer_diffusion(orc, gpu):
    """Same BC machinery on a non-trivial start: smooth velocity that vanishes on the walls, tracer blob (diffusivity 1e-2 so
    that the Crank-Nicolson scalar solve matters), x periodic, y slip / no-slip, z no-slip with moving lid."""
    n = (16, 16, 16)
    per = (1, 0, 0)
    x = [(np.arange(n[d]) + 0.5) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*x, indexing="ij")
    init = np.zeros(n + (5,))
    init[..., 0] = np.sin(2 * np.pi * X) * np.sin(np.pi * Y) * np.sin(np.pi * Z)
    init[..., 1] = 0.5 * np.cos(2 * np.pi * X) * np.sin(np.pi * Y) ** 2 * np.sin(2 * np.pi * Z)
    init[..., 2] = 0.3 * np.sin(4 * np.pi * X) * np.sin(np.pi * Y) * np.sin(np.pi * Z) ** 2
    init[..., 3] = 1.0 + 0.2 * np.sin(2 * np.pi * X) * np.cos(np.pi * Y)
    init[..., 4] = np.exp(-40.0 * ((X - 0.5) ** 2 + (Y - 0.4) ** 2 + (Z - 0.6) ** 2))
    kw = dict(cfl=0.5, visc_coef=0.02, init_iter=2, tracer_diff_coef=0.01)
    ref = run_oracle(orc, n, per, (0, 4, 5), (0, 5, 5), LID, 3, init, **kw)
    ns, lay, g, dts = run_gpu(gpu, n, per, (0, 4, 5), (0, 5, 5), LID, 3, init, 8, **kw)
    S = compare(gpu, ns, lay, g, n, dts, ref)
    # the tracer diffused: its maximum dropped markedly below the advected-only bound
    assert S[..., 4].max() < 0.97 * init[..., 4].max()
    # total tracer is conserved by diffusion with zero-flux walls up to the (non-conservative) advection: sanity only
    assert abs(S[..., 4].sum() - init[..., 4].sum()) < 0.05 * init[..., 4].sum()


def test_ragged_boxes_anisotropic_mesh_with_walls(orc, gpu):
    """non-cubic domain 48x32x16 with dx != dy != dz, chopped into boxes of unequal size (32+16 cells in x), x periodic,
    y slip / no-slip, z no-slip + lid, variable density, diffusive tracer: 2 pressure iterations + 3 steps vs the oracle"""
    n = (48, 32, 16)
    per = (1, 0, 0)
    prob_hi = (1.5, 1.0, 0.4)
    x = [(np.arange(n[d]) + 0.5) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*x, indexing="ij")
    init = np.zeros(n + (5,))
    init[..., 0] = 0.8 * np.sin(2 * np.pi * X) * np.sin(np.pi * Y) * np.sin(np.pi * Z)
    init[..., 1] = 0.3 * np.cos(4 * np.pi * X) * np.sin(np.pi * Y) ** 2 * np.sin(2 * np.pi * Z)
    init[..., 2] = 0.1 * np.sin(2 * np.pi * X) * np.sin(2 * np.pi * Y) * np.sin(np.pi * Z) ** 2
    init[..., 3] = 1.0 + 0.3 * np.cos(2 * np.pi * X) * np.sin(np.pi * Y)
    init[..., 4] = np.exp(-30.0 * ((X - 0.3) ** 2 + (Y - 0.6) ** 2 + (Z - 0.5) ** 2))
    kw = dict(cfl=0.4, visc_coef=0.005, init_iter=2, tracer_diff_coef=0.004)
    ref = run_oracle(orc, n, per, (0, 4, 5), (0, 5, 5), LID, 3, init, prob_hi=prob_hi, **kw)
    ns, lay, g, dts = run_gpu(gpu, n, per, (0, 4, 5), (0, 5, 5), LID, 3, init, (32, 16, 16), prob_hi=prob_hi, **kw)
    assert lay.nlocal() == 4
    compare(gpu, ns, lay, g, n, dts, ref)


def test_rejects_unsupported_physical_bc(gpu):
    from iamr_amd import ns as N
    lib = gpu
    n = (8, 8, 8)
    g = lib.Geom.make(n, periodic=(1, 1, 0))
    lay = lib.Layout.single(n)
    with pytest.raises(RuntimeError):
        N.NavierStokes(g, lay, N.ns_params(phys_lo=[0, 0, 6], phys_hi=[0, 0, 5]))                   # not a physical BC type
    N.NavierStokes(g, lay, N.ns_params(phys_lo=[0, 0, 1], phys_hi=[0, 0, 2], gravity=-1.0))         # outflow on top with gravity: hydrostatic pressure zero there
    bottom = N.NavierStokes(g, lay, N.ns_params(phys_lo=[0, 0, 2], phys_hi=[0, 0, 1], gravity=-1.0))
    bottom.init_rest(1.0)
    with pytest.raises(RuntimeError):                                                               # outflow at the bottom with gravity: Projection::computeRhoG aborts
        bottom.post_init(-1.0)
    N.NavierStokes(g, lay, N.ns_params(phys_lo=[0, 0, 1], phys_hi=[0, 0, 2]))                       # inflow / outflow are accepted
    N.NavierStokes(g, lay, N.ns_params(phys_lo=[0, 0, 3], phys_hi=[0, 0, 3]))                       # so is Symmetry


@pytest.mark.parametrize("kw", [
    dict(gravity=-9.8, use_forces_in_trans=0),
    dict(gravity=-9.8, use_forces_in_trans=1),
    dict(be_cn_theta=1.0),
    dict(init_iter=0, init_vel_iter=0, fixed_dt=2.0e-3),
    dict(visc_coef=0.0, tracer_diff_coef=0.0),
    dict(do_mom_diff=1, gravity=-9.8, use_forces_in_trans=1),
    dict(do_mom_diff=1, be_cn_theta=1.0),
    dict(do_cons_trac=1),
    dict(do_mom_diff=1, do_cons_trac=1, visc_coef=0.0, tracer_diff_coef=0.0, gravity=-9.8),
], ids=["gravity", "gravity+forces_in_trans", "backward_euler", "no_init_iters_fixed_dt", "inviscid",
        "mom_diff+gravity", "mom_diff+backward_euler", "cons_trac", "mom_diff+cons_trac_inviscid"])
def test_parameter_variants_match_oracle(orc, gpu, kw):
    """ns.* knobs that change the code path of the step (buoyancy forcing with variable density, godunov.use_forces_in_trans,
    be_cn_theta = 1, no initial iterations + fixed_dt, inviscid, ns.do_mom_diff = 1 (momentum-form velocity update, tensor solve with
    rho_flag 3), ns.do_cons_trac = 1 (conservative tracer, diffusion of S/rho with rho_flag 2)): periodic x, walls in y and z, 2 + 2 boxes, 3 steps"""
    n = (16, 16, 16)
    per = (1, 0, 0)
    x = [(np.arange(n[d]) + 0.5) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*x, indexing="ij")
    init = np.zeros(n + (5,))
    init[..., 0] = 0.6 * np.sin(2 * np.pi * X) * np.sin(np.pi * Y) * np.sin(np.pi * Z)
    init[..., 1] = 0.3 * np.cos(2 * np.pi * X) * np.sin(np.pi * Y) ** 2 * np.sin(2 * np.pi * Z)
    init[..., 2] = 0.2 * np.sin(4 * np.pi * X) * np.sin(np.pi * Y) * np.sin(np.pi * Z) ** 2
    init[..., 3] = 1.0 + 0.4 * np.exp(-25.0 * ((X - 0.5) ** 2 + (Y - 0.5) ** 2 + (Z - 0.6) ** 2))
    init[..., 4] = np.cos(2 * np.pi * X) * Y * (1 - Z)
    base = dict(cfl=0.5, visc_coef=0.01, init_iter=2, tracer_diff_coef=0.005)
    base.update(kw)
    nolid = [0.0] * 9
    ref = run_oracle(orc, n, per, (0, 4, 5), (0, 5, 4), nolid, 3, init, **base)
    ns, lay, g, dts = run_gpu(gpu, n, per, (0, 4, 5), (0, 5, 4), nolid, 3, init, (8, 16, 8), **base)
    compare(gpu, ns, lay, g, n, dts, ref)


def test_symmetry_planes_match_oracle(orc, gpu):
    """Symmetry (3) in y (normal velocity reflect_odd, everything else reflect_even; the tensor solve sees LinOpBCType::reflect_odd
    for v), no-slip z with the lid, x periodic: product against the oracle"""
    n = (16, 16, 16)
    per = (1, 0, 0)
    x = [(np.arange(n[d]) + 0.5) / n[d] for d in range(3)]
    X, Y, Z = np.meshgrid(*x, indexing="ij")
    init = np.zeros(n + (5,))
    init[..., 0] = np.sin(2 * np.pi * X) * np.cos(np.pi * Y) * np.sin(np.pi * Z)
    init[..., 1] = 0.5 * np.cos(2 * np.pi * X) * np.sin(np.pi * Y) * np.sin(2 * np.pi * Z)        # odd about the symmetry planes
    init[..., 2] = 0.3 * np.sin(4 * np.pi * X) * np.cos(np.pi * Y) * np.sin(np.pi * Z) ** 2
    init[..., 3] = 1.0 + 0.2 * np.sin(2 * np.pi * X) * np.cos(np.pi * Y)
    init[..., 4] = np.exp(-40.0 * ((X - 0.5) ** 2 + (Y - 0.4) ** 2 + (Z - 0.6) ** 2))
    kw = dict(cfl=0.5, visc_coef=0.02, init_iter=2, tracer_diff_coef=0.01)
    ref = run_oracle(orc, n, per, (0, 3, 5), (0, 3, 5), LID, 3, init, **kw)
    ns, lay, g, dts = run_gpu(gpu, n, per, (0, 3, 5), (0, 3, 5), LID, 3, init, 8, **kw)
    compare(gpu, ns, lay, g, n, dts, ref)


def test_symmetry_plane_reproduces_half_of_a_symmetric_periodic_flow(gpu):
    """pin of the Symmetry BC that needs no oracle: TaylorGreen is odd in u / even in v, w, rho about the planes x = 0 and x = 1/2,
    so the half domain [0, 1/2] with Symmetry faces in x carries the same solution as the periodic unit cube"""
    from iamr_amd import ns as N
    lib = gpu
    kw = dict(cfl=0.5, visc_coef=0.01, init_iter=2)
    gF = lib.Geom.make((32, 16, 16))
    nsF = N.NavierStokes(gF, lib.Layout.single((32, 16, 16)), N.ns_params(**kw))
    gH = lib.Geom.make((16, 16, 16), prob_hi=(0.5, 1.0, 1.0), periodic=(0, 1, 1))
    nsH = N.NavierStokes(gH, lib.Layout.single((16, 16, 16)), N.ns_params(phys_lo=[3, 0, 0], phys_hi=[3, 0, 0], **kw))
    for ns in (nsF, nsH):
        ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
        ns.post_init(-1.0)
    for _ in range(4):
        dF, dH = nsF.step(), nsH.step()
        assert abs(dF - dH) <= 1e-10 * dF
    SF = nsF.data(N.NavierStokes.S_NEW).gather_valid((32, 16, 16))
    SH = nsH.data(N.NavierStokes.S_NEW).gather_valid((16, 16, 16))
    assert np.abs(SH - SF[:16]).max() <= 1e-8
    assert np.abs(SF[..., 0]).max() > 0.5


@pytest.mark.parametrize("forms", [dict(), dict(do_mom_diff=1, do_cons_trac=1)], ids=["convective", "mom_diff+cons_trac"])
def test_rayleigh_taylor_matches_oracle(orc, gpu, forms):
    """BASELINE config C5 reduced to its single-level physics at a size the oracle finishes quickly: probtype 10 (tanh density /
    tracer interface with the reference's hard-coded perturbation), gravity, slip walls in z, periodic x / y, variable density in
    both projections, forces in the transverse terms (godunov.use_forces_in_trans = 1 as in regtest.3d.rayleightaylor); the second
    case uses that regtest's ns.do_mom_diff = 1 / ns.do_cons_trac = 1 (Exec/run3d/regtest.3d.rayleightaylor:6-7)"""
    from iamr_amd import ns as N
    lib = gpu
    L = orc.lib()
    n, per, prob_lo, prob_hi = (16, 16, 32), (1, 1, 0), (0.0, 0.0, 0.0), (1.0, 1.0, 2.0)
    kw = dict(cfl=0.7, visc_coef=0.0, init_iter=3, gravity=-9.8, use_forces_in_trans=1, **forms)
    rt = dict(rho_1=1.0, rho_2=2.0, tra_1=1.0, tra_2=0.0, pertamp=0.1, interface_width=0.08)
    # oracle
    g_o = orc.geom(n, problo=prob_lo, probhi=prob_hi, periodic=per)
    p = oracle_params(orc, per, (0, 0, 4), (0, 0, 4), [0.0] * 9, **kw)
    o = orc.mg_opts()
    s = C.c_void_p(L.orc_ns_create(C.byref(g_o), C.byref(p), C.byref(o)))
    L.orc_ns_init_rayleightaylor(s, *[C.c_double(rt[k]) for k in ("rho_1", "rho_2", "tra_1", "tra_2", "pertamp", "interface_width")])
    L.orc_ns_post_init(s, C.c_double(-1.0))
    dts_o = [L.orc_ns_step(s) for _ in range(3)]
    S_o = orc.from_cfab(L.orc_ns_fab(s, 0)).valid(n).copy()
    L.orc_ns_destroy(s)
    # product, two boxes in z
    g_d = lib.Geom.make(n, prob_lo=prob_lo, prob_hi=prob_hi, periodic=per)
    lay = lib.Layout.decompose(n, 16)
    ns = N.NavierStokes(g_d, lay, N.ns_params(phys_lo=[0, 0, 4], phys_hi=[0, 0, 4], **kw))
    ns.init_rayleightaylor(**rt)
    ns.post_init(-1.0)
    dts = [ns.step() for _ in range(3)]
    assert np.allclose(dts, dts_o, rtol=1e-9, atol=0)
    S = ns.data(N.NavierStokes.S_NEW).gather_valid(n)
    for comp in range(5):
        scale = max(np.abs(S_o[..., comp]).max(), 1e-3)
        assert np.abs(S[..., comp] - S_o[..., comp]).max() <= 1e-8 * scale, comp
    assert np.abs(S[..., 2]).max() > 1e-3 and S[..., 3].min() > 0.99 and S[..., 3].max() < 2.01     # the heavy fluid starts to fall


def test_euler_regtest_single_level_matches_oracle(orc, gpu):
    """Exec/run3d/regtest.3d.euler reduced to its base level (32^3, periodic, inviscid, cfl 0.9, prob.probtype = 7: vortex tube with a
    wobble, Source/prob/prob_init.cpp:562-610): initial data from iamr_amd/probinit.py on both sides, three steps; and the same run
    through the inputs-file driver (iamr_amd.run.build)"""
    from iamr_amd.probinit import cell_centres, initial_state
    from iamr_amd.inputs import Inputs
    from iamr_amd import run as R, ns as N
    n, per = (32, 32, 32), (1, 1, 1)
    prob = dict(probtype=7, density_ic=1.0)
    init = initial_state(prob, *cell_centres(n, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)))
    kw = dict(cfl=0.9, visc_coef=0.0, tracer_diff_coef=0.0, init_shrink=1.0)
    nolid = [0.0] * 9
    ref = run_oracle(orc, n, per, (0, 0, 0), (0, 0, 0), nolid, 3, init, **kw)
    ns, lay, g, dts = run_gpu(gpu, n, per, (0, 0, 0), (0, 0, 0), nolid, 3, init, (16, 32, 16), **kw)
    compare(gpu, ns, lay, g, n, dts, ref)
    S1 = ns.data(N.NavierStokes.S_NEW).gather_valid(n)
    ldc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "inputs.3d.lid_driven_cavity16")
    inp = Inputs([ldc], ["amr.n_cell=32 32 32", "amr.max_grid_size=16", "geometry.is_periodic=1 1 1", "ns.lo_bc=0 0 0", "ns.hi_bc=0 0 0",
                         "prob.probtype=7", "ns.cfl=0.9", "ns.init_shrink=1.0", "ns.vel_visc_coef=0.0", "ns.scal_diff_coefs=0.0",
                         "ns.init_iter=2", "ns.init_dt=-1.0"])
    ns2, lay2, g2, pr = R.build(inp, gpu, N)
    ns2.post_init(-1.0)
    dts2 = [ns2.step() for _ in range(3)]
    S2 = ns2.data(N.NavierStokes.S_NEW).gather_valid(n)
    assert np.allclose(dts2, dts, rtol=1e-10, atol=0) and np.abs(S2 - S1).max() < 1e-9
    assert np.abs(S1[..., 2]).max() > 0.04 and np.abs(S1[..., 0]).max() > 0.99


def test_scal_min_max_matches_oracle_and_bounds_the_scalars(orc, gpu):
    """ns.do_denminmax / ns.do_scalminmax (Source/NavierStokesBase.cpp:466-467): ConservativeScalMinMax / ConvectiveScalMinMax
    (:4256-4368) after the advective update -- a sharp tracer / density front advected by the Taylor-Green flow: product vs oracle, and
    the clipped fields stay inside the initial bounds where the unclipped run overshoots"""
    from iamr_amd import ns as N
    lib, L = gpu, orc.lib()
    n = (16, 16, 16)
    res = {}
    for clip in (0, 1):
        kw = dict(cfl=0.9, visc_coef=0.0, init_iter=1, do_denminmax=clip, do_scalminmax=clip)
        g_o = orc.geom(n)
        s = C.c_void_p(L.orc_ns_create(C.byref(g_o), C.byref(orc.ns_params(**kw)), C.byref(orc.mg_opts())))
        So = orc.from_cfab(L.orc_ns_fab(s, 0))
        X, Y, Z = np.meshgrid(*[(np.arange(n[d]) + 0.5) / n[d] for d in range(3)], indexing="ij")
        S0 = orc.taylorgreen_state(X, Y, Z, c=1.0)
        S0[..., 3] = np.where(np.abs(X - 0.5) < 0.2, 2.0, 1.0)            # density step
        S0[..., 4] = np.where((np.abs(Y - 0.5) < 0.2) & (np.abs(Z - 0.5) < 0.25), 1.0, 0.0)   # tracer box
        So.valid(n)[...] = S0
        L.orc_ns_post_init(s, C.c_double(-1.0))
        dts_o = [L.orc_ns_step(s) for _ in range(3)]
        ref = orc.from_cfab(L.orc_ns_fab(s, 0)).valid(n).copy()
        L.orc_ns_destroy(s)
        g_d = lib.Geom.make(n)
        lay = lib.Layout.decompose(n, 8)
        ns = N.NavierStokes(g_d, lay, N.ns_params(**kw))
        m = lib.MultiFab(lay, lib.CELL, 5, 1)
        G = np.zeros(tuple(v + 2 for v in n) + (5,), order="F")
        G[1:-1, 1:-1, 1:-1] = S0
        m.set_from_global(G, (-1, -1, -1))
        ns.set_data(ns.S_NEW, m)
        ns.post_init(-1.0)
        dts = [ns.step() for _ in range(3)]
        got = ns.data(ns.S_NEW).gather_valid(n)
        assert np.allclose(dts, dts_o, rtol=1e-9, atol=0)
        for c in range(5):
            assert np.abs(got[..., c] - ref[..., c]).max() <= 2e-8 * max(1.0, np.abs(ref[..., c]).max()), (clip, c)
        res[clip] = got
    # density: upstream calls ConservativeScalMinMax(S_new, Density, Density, ...) -- new rho / new rho = 1 clipped to the range of
    # old rho / old rho = 1 and multiplied back: a no-op as written (NavierStokesBase.cpp:2771-2788), followed as written
    assert np.array_equal(res[1][..., 3] == res[0][..., 3], np.ones(n, bool)) or np.abs(res[1][..., 3] - res[0][..., 3]).max() < 1e-6
    assert res[1][..., 4].min() >= -1e-12 and res[1][..., 4].max() <= 1.0 + 1e-12
    assert res[0][..., 4].max() > 1.0 + 1e-4 or res[0][..., 4].min() < -1e-4        # the unclipped run does overshoot
