"""Python mirror of the operator layer for tests / bench: nodal projection, tensor diffusion and the
NavierStokes level object, all thin wrappers over the C-ABI (include/iamrx.h)."""
import ctypes as C
from .lib import lib, check, i3, mg_opts, MgStats, MultiFab, _h


class NsParams(C.Structure):
    _fields_ = [("cfl", C.c_double), ("visc_coef", C.c_double), ("be_cn_theta", C.c_double), ("gravity", C.c_double),
                ("mac_tol", C.c_double), ("mac_abs_tol", C.c_double), ("proj_tol", C.c_double),
                ("proj_abs_tol", C.c_double), ("visc_tol", C.c_double),
                ("use_forces_in_trans", C.c_int), ("do_mom_diff", C.c_int), ("init_iter", C.c_int),
                ("init_vel_iter", C.c_int), ("init_shrink", C.c_double), ("change_max", C.c_double),
                ("fixed_dt", C.c_double), ("nscal", C.c_int), ("verbose", C.c_int),
                ("init_dt", C.c_double), ("tracer_diff_coef", C.c_double), ("phys_lo", C.c_int * 3), ("phys_hi", C.c_int * 3),
                ("wall_vel_lo", C.c_double * 9), ("wall_vel_hi", C.c_double * 9),
                ("scal_bc_lo", C.c_double * 12), ("scal_bc_hi", C.c_double * 12), ("do_cons_trac", C.c_int), ("do_denminmax", C.c_int), ("do_scalminmax", C.c_int),
                ("do_trac2", C.c_int), ("do_cons_trac2", C.c_int), ("tracer2_diff_coef", C.c_double), ("do_temp", C.c_int), ("temp_cond_coef", C.c_double), ("use_ppm", C.c_int)]


def ns_params(**kw):
    p = NsParams()
    lib().iamrx_ns_default_params(C.byref(p))
    for k, v in kw.items():
        if k in ("phys_lo", "phys_hi"):
            setattr(p, k, (C.c_int * 3)(*[int(x) for x in v]))
        elif k in ("wall_vel_lo", "wall_vel_hi"):
            setattr(p, k, (C.c_double * 9)(*[float(x) for x in v]))
        elif k in ("scal_bc_lo", "scal_bc_hi"):        # [d*4+n]; a 6-entry list is the two-scalar layout [d*2+n]
            v = [float(x) for x in v]
            if len(v) == 6:
                v = [v[2 * d + n] if n < 2 else 0.0 for d in range(3) for n in range(4)]
            setattr(p, k, (C.c_double * 12)(*v))
        else:
            setattr(p, k, v)
    return p


def nodal_residual(geom, out, phi, sig, rhs=None):
    check(lib().iamrx_nodal_residual(C.byref(geom), out.h, phi.h, sig.h, _h(rhs)))


def nodal_gs_color(geom, phi, rhs, sig, color):
    check(lib().iamrx_nodal_gs_color(C.byref(geom), phi.h, rhs.h, sig.h, color))


def nodal_gs_sweep(geom, phi, rhs, sig, fused=1):
    """one 8-colour Gauss-Seidel sweep incl. ghost fills (fused: plane-fused two-pass variant, same arithmetic)"""
    check(lib().iamrx_nodal_gs_sweep(C.byref(geom), phi.h, rhs.h, sig.h, int(fused)))


def nodal_restrict(crse, fine):
    check(lib().iamrx_nodal_restrict(crse.h, fine.h))


def nodal_interp_add(fine, crse, sig_fine):
    check(lib().iamrx_nodal_interp_add(fine.h, crse.h, sig_fine.h))


def nodal_divu(geom, rhs, vel, vcomp=0):
    check(lib().iamrx_nodal_divu(C.byref(geom), rhs.h, vel.h, vcomp))


def nodal_compgrad(geom, gp, phi):
    """MLNodeLaplacian::compGrad (NavierStokesBase::computeGradP, reference Source/NavierStokesBase.cpp:4102-4122)"""
    check(lib().iamrx_nodal_compgrad(C.byref(geom), gp.h, phi.h))


def nodal_solve(geom, phi, rhs, sig, sig_comp=0, lobc=(0, 0, 0), hibc=(0, 0, 0), rel_tol=1e-12, abs_tol=1e-16, opts=None):
    """div(sig grad phi) = rhs; Dirichlet nodes (outflow faces, boundary of a level that does not cover the domain) keep phi"""
    st = MgStats()
    o = opts if opts is not None else mg_opts()
    check(lib().iamrx_nodal_solve(C.byref(geom), _h(phi), _h(rhs), _h(sig), sig_comp, i3(lobc), i3(hibc),
                                      C.c_double(rel_tol), C.c_double(abs_tol), C.byref(o), C.byref(st)))
    return st


def nodal_projection(geom, vel, vcomp, phi, sig, sig_comp=0, lobc=(0, 0, 0), hibc=(0, 0, 0), rel_tol=1e-12, abs_tol=1e-16,
                     opts=None, gp=None, increment_gp=False):
    """Projection::doMLMGNodalProjection on one level (reference Source/Projection.cpp:2385-2567)"""
    st = MgStats()
    o = opts if opts is not None else mg_opts()
    check(lib().iamrx_nodal_projection(C.byref(geom), vel.h, vcomp, phi.h, sig.h, sig_comp, i3(lobc), i3(hibc),
                                       C.c_double(rel_tol), C.c_double(abs_tol), C.byref(o), _h(gp), int(increment_gp), C.byref(st)))
    return st


def _bcn(lobc, hibc):
    """(lo, hi, nbc): lobc/hibc are 3 codes, or 3 x 3 codes (one triple per velocity component)"""
    if hasattr(lobc[0], "__len__"):
        flat_lo = [int(v) for c in lobc for v in c]
        flat_hi = [int(v) for c in hibc for v in c]
        return (C.c_int * 9)(*flat_lo), (C.c_int * 9)(*flat_hi), 3
    return i3(lobc), i3(hibc), 1


def tensor_apply(geom, out, vel, a, b, acoef, eta, lobc=(0, 0, 0), hibc=(0, 0, 0), maxorder=2):
    lo, hi, nbc = _bcn(lobc, hibc)
    check(lib().iamrx_tensor_apply(C.byref(geom), out.h, vel.h, C.c_double(a), C.c_double(b), _h(acoef), eta[0].h, eta[1].h,
                                   eta[2].h, lo, hi, nbc, maxorder))


def tensor_solve(geom, soln, rhs, a, b, acoef, eta, lobc=(0, 0, 0), hibc=(0, 0, 0), tol_rel=1e-10, tol_abs=0.0, opts=None):
    st = MgStats()
    o = opts if opts is not None else mg_opts(maxorder=2)
    lo, hi, nbc = _bcn(lobc, hibc)
    check(lib().iamrx_tensor_solve(C.byref(geom), soln.h, rhs.h, C.c_double(a), C.c_double(b), _h(acoef), eta[0].h, eta[1].h,
                                   eta[2].h, lo, hi, nbc, C.c_double(tol_rel), C.c_double(tol_abs), C.byref(o), C.byref(st)))
    return st


def tensor_apply_cf(geom, out, vel, a, b, acoef, eta, crse_vel, cgeom, ratio=2, lobc=(0, 0, 0), hibc=(0, 0, 0), maxorder=2):
    """tensor operator on a refined level (tensorop.setCoarseFineBC(&crsedata, ratio), Source/Diffusion.cpp:1725-1736); crse_vel None: homogeneous"""
    lo, hi, nbc = _bcn(lobc, hibc)
    check(lib().iamrx_tensor_apply_cf(C.byref(geom), out.h, vel.h, C.c_double(a), C.c_double(b), _h(acoef), eta[0].h, eta[1].h,
                                      eta[2].h, lo, hi, nbc, maxorder, _h(crse_vel), C.byref(cgeom), int(ratio)))


def tensor_solve_cf(geom, soln, rhs, a, b, acoef, eta, crse_vel, cgeom, ratio=2, lobc=(0, 0, 0), hibc=(0, 0, 0), tol_rel=1e-10, tol_abs=0.0,
                    opts=None):
    st = MgStats()
    o = opts if opts is not None else mg_opts(maxorder=2)
    lo, hi, nbc = _bcn(lobc, hibc)
    check(lib().iamrx_tensor_solve_cf(C.byref(geom), soln.h, rhs.h, C.c_double(a), C.c_double(b), _h(acoef), eta[0].h, eta[1].h,
                                      eta[2].h, lo, hi, nbc, _h(crse_vel), C.byref(cgeom), int(ratio), C.c_double(tol_rel), C.c_double(tol_abs),
                                      C.byref(o), C.byref(st)))
    return st


class NavierStokes:
    """level object with the reference's method names (NavierStokes::advance, post_init, ...)"""
    S_NEW, S_OLD, P_NEW, P_OLD, GP_NEW, GP_OLD, UMAC_X, UMAC_Y, UMAC_Z, AOFS = range(10)
    _types = {0: ((0, 0, 0), 5, 1), 1: ((0, 0, 0), 5, 1), 2: ((1, 1, 1), 1, 1), 3: ((1, 1, 1), 1, 1), 4: ((0, 0, 0), 3, 1),
              5: ((0, 0, 0), 3, 1), 6: ((1, 0, 0), 1, 1), 7: ((0, 1, 0), 1, 1), 8: ((0, 0, 1), 1, 1), 9: ((0, 0, 0), 5, 0)}

    def __init__(self, geom, layout, params=None, opts=None):
        self.geom = geom
        self.layout = layout
        self.params = params if params is not None else ns_params()
        self.opts = opts if opts is not None else mg_opts()
        self.h = C.c_void_p()
        check(lib().iamrx_ns_create(C.byref(geom), layout.h, C.byref(self.params), C.byref(self.opts), C.byref(self.h)))

    def init_rayleightaylor(self, rho_1, rho_2, tra_1=0.0, tra_2=0.0, pertamp=0.0, interface_width=1.0):
        check(lib().iamrx_ns_init_rayleightaylor(self.h, C.c_double(rho_1), C.c_double(rho_2), C.c_double(tra_1), C.c_double(tra_2),
                                                 C.c_double(pertamp), C.c_double(interface_width)))

    def init_taylorgreen(self, vfac=1.0, a=1.0, b=1.0, c=0.0, rho0=1.0):
        check(lib().iamrx_ns_init_taylorgreen(self.h, C.c_double(vfac), C.c_double(a), C.c_double(b), C.c_double(c), C.c_double(rho0)))

    def init_rest(self, rho0=1.0):
        check(lib().iamrx_ns_init_rest(self.h, C.c_double(rho0)))

    def set_data(self, which, mf):
        check(lib().iamrx_ns_set_data(self.h, int(which), mf.h))

    def post_init(self, stop_time=-1.0):
        check(lib().iamrx_ns_post_init(self.h, C.c_double(stop_time)))

    def step(self):
        dt = C.c_double()
        check(lib().iamrx_ns_step(self.h, C.byref(dt)))
        return dt.value

    def advance(self, dt):
        est = C.c_double()
        check(lib().iamrx_ns_advance(self.h, C.c_double(dt), C.byref(est)))
        return est.value

    @property
    def time(self):
        t = C.c_double()
        check(lib().iamrx_ns_time(self.h, C.byref(t), None, None))
        return t.value

    @property
    def dt(self):
        t = C.c_double()
        check(lib().iamrx_ns_time(self.h, None, C.byref(t), None))
        return t.value

    def data(self, which):
        h = C.c_void_p()
        check(lib().iamrx_ns_data(self.h, which, C.byref(h)))
        typ, nc, ng = self._types[which]
        if nc == 5:
            nc = self.nalloc if which in (0, 1) else self.nstate
        return MultiFab(self.layout, typ, nc, ng, _handle=h, _owned=True)

    def derive(self, name):
        """derived quantity of the plotfile ("energy", "mag_vort", "avg_pressure": NavierStokes::derive) as a one-component cell MultiFab"""
        from .lib import MultiFab, CELL
        out = MultiFab(self.layout, CELL, 1, 0)
        check(lib().iamrx_ns_derive(self.h, name.encode(), out.h, 0))
        return out

    @property
    def nstate(self):
        """NUM_STATE (NavierStokes.cpp:43-48): u v w density tracer [tracer2] [temp]"""
        return 5 + (1 if self.params.do_trac2 else 0) + (1 if self.params.do_temp else 0)

    @property
    def nalloc(self):
        """components of the S arrays: the state, plus divu and dsdt (Divu_Type, Dsdt_Type) in a temperature run"""
        return self.nstate + (2 if self.params.do_temp else 0)

    def stats(self):
        a, b, c = MgStats(), MgStats(), MgStats()
        check(lib().iamrx_ns_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return a, b, c

    def profile(self, enable):
        arr = (C.c_double * 8)()
        check(lib().iamrx_ns_profile(self.h, int(enable), arr))
        return list(arr)

    def __del__(self):
        try:
            if self.h:
                lib().iamrx_ns_destroy(self.h)
        except Exception:
            pass
