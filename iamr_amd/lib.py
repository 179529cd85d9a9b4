"""ctypes binding of libiamrx.so (C-ABI declared in include/iamrx.h).

The HIP library is the product; there is no CPU fallback: importing works without a GPU (so that the
build / symbol checks can run), but `init()` raises if no gfx950 device is present.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libiamrx.so")
_lib = None


class IamrxError(RuntimeError):
    pass


class Geom(C.Structure):
    _fields_ = [("dom_lo", C.c_int * 3), ("dom_hi", C.c_int * 3), ("prob_lo", C.c_double * 3),
                ("prob_hi", C.c_double * 3), ("periodic", C.c_int * 3)]

    @staticmethod
    def make(n, prob_lo=(0.0, 0.0, 0.0), prob_hi=(1.0, 1.0, 1.0), periodic=(1, 1, 1)):
        g = Geom()
        g.dom_lo = (C.c_int * 3)(0, 0, 0)
        g.dom_hi = (C.c_int * 3)(*[int(n[d]) - 1 for d in range(3)])
        g.prob_lo = (C.c_double * 3)(*prob_lo)
        g.prob_hi = (C.c_double * 3)(*prob_hi)
        g.periodic = (C.c_int * 3)(*periodic)
        return g

    @property
    def n(self):
        return tuple(self.dom_hi[d] - self.dom_lo[d] + 1 for d in range(3))

    @property
    def dx(self):
        return tuple((self.prob_hi[d] - self.prob_lo[d]) / self.n[d] for d in range(3))


class MgOpts(C.Structure):
    _fields_ = [("nu1", C.c_int), ("nu2", C.c_int), ("nuf", C.c_int), ("nub", C.c_int), ("max_iters", C.c_int),
                ("bottom_maxiter", C.c_int), ("bottom_reltol", C.c_double), ("omega", C.c_double),
                ("maxorder", C.c_int), ("max_coarsening_level", C.c_int), ("min_width", C.c_int),
                ("nodal_sweeps", C.c_int), ("nodal_smoother", C.c_int), ("verbose", C.c_int),
                ("bottom_smoother_only", C.c_int), ("fixed_iters", C.c_int), ("nodal_nu1", C.c_int), ("nodal_nu2", C.c_int),
                ("device_bottom", C.c_int), ("slab", C.c_int)]


class MgStats(C.Structure):
    _fields_ = [("iters", C.c_int), ("resnorm0", C.c_double), ("rhsnorm0", C.c_double), ("resnorm", C.c_double),
                ("bottom_iters_total", C.c_int), ("converged", C.c_int), ("vcycle_ms", C.c_double), ("nlevels", C.c_int)]


def lib():
    """load libiamrx.so (fails loudly if it has not been built)"""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IamrxError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(there is no CPU fallback for the product path)")
        _lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        _lib.iamrx_last_error.restype = C.c_char_p
        _lib.iamrx_stream.restype = C.c_void_p
    return _lib


def check(rc):
    if rc != 0:
        raise IamrxError(lib().iamrx_last_error().decode())


_initialized = False


def init(device=0):
    global _initialized
    check(lib().iamrx_init(int(device)))
    _initialized = True


def tuning_set(key, value):
    """run-time switch of the library (include/iamrx.h: iamrx_tuning_set); key without the IAMRX_ prefix"""
    check(lib().iamrx_tuning_set(key.encode(), C.c_double(float(value))))


def tuning_get(key, default):
    v = C.c_double()
    check(lib().iamrx_tuning_get(key.encode(), C.c_double(float(default)), C.byref(v)))
    return v.value


def sync():
    check(lib().iamrx_sync())


def mg_opts(**kw):
    o = MgOpts()
    lib().iamrx_mg_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


CELL = (0, 0, 0)
NODE = (1, 1, 1)


def face(d):
    t = [0, 0, 0]
    t[d] = 1
    return tuple(t)


class Layout:
    """BoxArray + DistributionMapping of one level."""

    def __init__(self, boxes, owners=None):
        nb = len(boxes)
        arr = (C.c_int * (6 * nb))()
        for i, (lo, hi) in enumerate(boxes):
            for d in range(3):
                arr[6 * i + d] = int(lo[d])
                arr[6 * i + 3 + d] = int(hi[d])
        own = (C.c_int * nb)(*([0] * nb if owners is None else [int(o) for o in owners]))
        self.h = C.c_void_p()
        check(lib().iamrx_layout_create(nb, arr, own, C.byref(self.h)))
        self.boxes = [(tuple(lo), tuple(hi)) for lo, hi in boxes]
        self.owners = [0] * nb if owners is None else list(owners)

    @staticmethod
    def from_handle(h, boxes, owners=None):
        """wrap a layout handle obtained from the library (e.g. iamrx_amr_level_layout)"""
        self = Layout.__new__(Layout)
        self.h = h
        self.boxes = [(tuple(lo), tuple(hi)) for lo, hi in boxes]
        self.owners = [0] * len(boxes) if owners is None else list(owners)
        return self

    @staticmethod
    def single(n):
        return Layout([((0, 0, 0), (n[0] - 1, n[1] - 1, n[2] - 1))])

    @staticmethod
    def decompose(n, max_grid_size, nranks=1):
        """chop the domain [0,n-1]^3 into boxes of at most max_grid_size, round-robin over ranks"""
        boxes = []
        mg = max_grid_size if hasattr(max_grid_size, "__len__") else (max_grid_size,) * 3
        for k0 in range(0, n[2], mg[2]):
            for j0 in range(0, n[1], mg[1]):
                for i0 in range(0, n[0], mg[0]):
                    boxes.append(((i0, j0, k0), (min(i0 + mg[0], n[0]) - 1, min(j0 + mg[1], n[1]) - 1, min(k0 + mg[2], n[2]) - 1)))
        per = (len(boxes) + nranks - 1) // nranks
        owners = [min(i // per, nranks - 1) for i in range(len(boxes))]
        return Layout(boxes, owners)

    def nlocal(self):
        n = C.c_int()
        check(lib().iamrx_layout_nlocal(self.h, C.byref(n)))
        return n.value

    def local_box(self, li):
        a = (C.c_int * 6)()
        gi = C.c_int()
        check(lib().iamrx_layout_local_box(self.h, li, a, C.byref(gi)))
        return tuple(a[0:3]), tuple(a[3:6]), gi.value

    def __del__(self):
        try:
            if self.h:
                lib().iamrx_layout_destroy(self.h)
        except Exception:
            pass


class MultiFab:
    """device-resident multi-component array over the boxes of a Layout"""

    def __init__(self, layout, typ=CELL, ncomp=1, ngrow=0, _handle=None, _owned=True):
        self.layout = layout
        self.typ = tuple(typ)
        self.ncomp = ncomp
        self.ngrow = ngrow
        self._owned = _owned
        if _handle is not None:
            self.h = _handle
        else:
            self.h = C.c_void_p()
            check(lib().iamrx_mf_create(layout.h, i3(typ), ncomp, ngrow, C.byref(self.h)))

    def nlocal(self):
        n = C.c_int()
        check(lib().iamrx_mf_info(self.h, None, None, None, C.byref(n)))
        return n.value

    def fab_box(self, li=0):
        a = (C.c_int * 6)()
        check(lib().iamrx_mf_fab_box(self.h, li, a))
        return tuple(a[0:3]), tuple(a[3:6])

    def to_numpy(self, li=0):
        """(nx,ny,nz,nc) Fortran-ordered copy of local fab li, ghost cells included; returns (array, lo)"""
        lo, hi = self.fab_box(li)
        shape = tuple(hi[d] - lo[d] + 1 for d in range(3)) + (self.ncomp,)
        a = np.empty(shape, dtype=np.float64, order="F")
        check(lib().iamrx_mf_to_host(self.h, li, a.ctypes.data_as(C.POINTER(C.c_double))))
        return a, lo

    def from_numpy(self, a, li=0):
        lo, hi = self.fab_box(li)
        shape = tuple(hi[d] - lo[d] + 1 for d in range(3)) + (self.ncomp,)
        a = np.asfortranarray(a, dtype=np.float64).reshape(shape, order="F")
        check(lib().iamrx_mf_from_host(self.h, li, a.ctypes.data_as(C.POINTER(C.c_double))))

    def set_from_global(self, G, glo):
        """fill every local fab (valid + ghosts) from a global numpy array G whose index origin is glo"""
        for li in range(self.nlocal()):
            lo, hi = self.fab_box(li)
            sl = tuple(slice(lo[d] - glo[d], hi[d] - glo[d] + 1) for d in range(3))
            self.from_numpy(G[sl], li)

    def gather_valid(self, n):
        """assemble the valid regions of all LOCAL fabs into a global array over cells [0,n-1] (+type)"""
        shape = tuple(n[d] + self.typ[d] for d in range(3)) + (self.ncomp,)
        G = np.zeros(shape, order="F")
        for li in range(self.nlocal()):
            a, lo = self.to_numpy(li)
            blo, bhi, _ = self.layout.local_box(li)
            vlo = blo
            vhi = tuple(bhi[d] + self.typ[d] for d in range(3))
            src = tuple(slice(vlo[d] - lo[d], vhi[d] - lo[d] + 1) for d in range(3))
            dst = tuple(slice(vlo[d], vhi[d] + 1) for d in range(3))
            G[dst] = a[src]
        return G

    def setval(self, v):
        check(lib().iamrx_mf_setval(self.h, C.c_double(v)))

    def fill_boundary(self, geom):
        check(lib().iamrx_mf_fill_boundary(self.h, C.byref(geom)))

    def fill_physbc(self, geom, bc, extdir_lo=None, extdir_hi=None, scomp=0, ncomp=None):
        """physical BC fill outside the domain; bc: list of (lo[3], hi[3]) BCType codes per component"""
        nc = self.ncomp if ncomp is None else ncomp
        el = None if extdir_lo is None else (C.c_double * (3 * nc))(*[float(x) for row in extdir_lo for x in row])
        eh = None if extdir_hi is None else (C.c_double * (3 * nc))(*[float(x) for row in extdir_hi for x in row])
        check(lib().iamrx_mf_fill_physbc(self.h, C.byref(geom), scomp, nc, _bcrec(nc, bc), el, eh))

    def norm0(self, comp=0, ncomp=None, ngrow=0):
        out = C.c_double()
        check(lib().iamrx_mf_norm0(self.h, comp, self.ncomp if ncomp is None else ncomp, ngrow, C.byref(out)))
        return out.value

    def dev_ptr(self, li=0):
        p = C.POINTER(C.c_double)()
        check(lib().iamrx_mf_dev_ptr(self.h, li, C.byref(p)))
        return C.cast(p, C.c_void_p).value

    def __del__(self):
        try:
            if self._owned and self.h:
                lib().iamrx_mf_destroy(self.h)
        except Exception:
            pass


def _h(m):
    return m.h if m is not None else None


def abec_gsrb(geom, alpha, beta, a, b, phi, rhs, redblack, omega=1.15, lobc=(0, 0, 0), hibc=(0, 0, 0), maxorder=3):
    check(lib().iamrx_abec_gsrb(C.byref(geom), C.c_double(alpha), C.c_double(beta), _h(a), b[0].h, b[1].h, b[2].h,
                                phi.h, rhs.h, redblack, C.c_double(omega), i3(lobc), i3(hibc), maxorder))


def parallel_copy(dst, src, scomp=0, dcomp=0, ncomp=None, src_ng=0, dst_ng=0, periodic_geom=None):
    nc = src.ncomp if ncomp is None else ncomp
    check(lib().iamrx_parallel_copy(dst.h, src.h, scomp, dcomp, nc, src_ng, dst_ng, C.byref(periodic_geom) if periodic_geom is not None else None))


def fillpatch_two_levels(dst, time, fine, crse, cgeom, fgeom, scomp=0, ncomp=None, dcomp=0, ratio=2, bc=None, extdir_lo=None, extdir_hi=None):
    """fine / crse: (old MultiFab or None, new MultiFab, t_old, t_new)"""
    nc = dst.ncomp if ncomp is None else ncomp
    el = None if extdir_lo is None else (C.c_double * (3 * nc))(*[float(x) for row in extdir_lo for x in row])
    eh = None if extdir_hi is None else (C.c_double * (3 * nc))(*[float(x) for row in extdir_hi for x in row])
    bcs = bc if bc is not None else [((0, 0, 0), (0, 0, 0))] * nc
    check(lib().iamrx_fillpatch_two_levels(dst.h, dcomp, C.c_double(time), _h(fine[0]), fine[1].h, C.c_double(fine[2]), C.c_double(fine[3]),
                                           _h(crse[0]), crse[1].h, C.c_double(crse[2]), C.c_double(crse[3]), scomp, nc,
                                           C.byref(cgeom), C.byref(fgeom), ratio, _bcrec(nc, bcs), el, eh))


def create_umac_grown(umac_fine, umac_crse, cgeom, fgeom, ratio=2, divu=None):
    check(lib().iamrx_create_umac_grown(umac_fine[0].h, umac_fine[1].h, umac_fine[2].h, umac_crse[0].h, umac_crse[1].h, umac_crse[2].h,
                                        _h(divu), C.byref(cgeom), C.byref(fgeom), ratio))


class FluxRegister:
    """amrex::FluxRegister role (names as in the reference: CrseInit, FineAdd, Reflux)"""

    def __init__(self, fine_layout, crse_layout, cgeom, ratio, ncomp):
        self.h = C.c_void_p()
        self._keep = (fine_layout, crse_layout)
        check(lib().iamrx_fluxreg_create(fine_layout.h, crse_layout.h, C.byref(cgeom), ratio, ncomp, C.byref(self.h)))

    def setVal(self, v):
        check(lib().iamrx_fluxreg_setval(self.h, C.c_double(v)))

    def CrseInit(self, flux, dir, scomp, dcomp, ncomp, mult, add=False):
        check(lib().iamrx_fluxreg_crse_init(self.h, flux.h, dir, scomp, dcomp, ncomp, C.c_double(mult), int(add)))

    def FineAdd(self, flux, dir, scomp, dcomp, ncomp, mult):
        check(lib().iamrx_fluxreg_fine_add(self.h, flux.h, dir, scomp, dcomp, ncomp, C.c_double(mult)))

    def Reflux(self, S, volume, scale, scomp, dcomp, ncomp):
        check(lib().iamrx_fluxreg_reflux(self.h, S.h, C.c_double(volume), C.c_double(scale), scomp, dcomp, ncomp))

    def __del__(self):
        try:
            if self.h:
                lib().iamrx_fluxreg_destroy(self.h)
                self.h = None
        except Exception:
            pass


class SyncRegister:
    """IAMR's SyncRegister (Source/SyncRegister.H:10-66; names as in the reference: CrseInit, FineAdd, InitRHS) of one coarse/fine interface"""

    def __init__(self, fine_layout, crse_layout, cgeom, fgeom, ratio=2, phys_lo=(0, 0, 0), phys_hi=(0, 0, 0)):
        self.h = C.c_void_p()
        self._keep = (fine_layout, crse_layout)
        check(lib().iamrx_syncreg_create(fine_layout.h, crse_layout.h, C.byref(cgeom), C.byref(fgeom), int(ratio), i3(phys_lo), i3(phys_hi), C.byref(self.h)))

    def CrseInit(self, sync_resid_crse, mult):
        check(lib().iamrx_syncreg_crse_init(self.h, sync_resid_crse.h, C.c_double(mult)))

    def FineAdd(self, sync_resid_fine, mult):
        check(lib().iamrx_syncreg_fine_add(self.h, sync_resid_fine.h, C.c_double(mult)))

    def CompAdd(self, sync_resid_fine, fgeom, finer_layout, finer_ratio, mult):
        """Source/SyncRegister.cpp:302-348: zero the residual on the nodes of the next finer level's boxes, then FineAdd (modifies the residual)"""
        check(lib().iamrx_syncreg_comp_add(self.h, sync_resid_fine.h, C.byref(fgeom), finer_layout.h, int(finer_ratio), C.c_double(mult)))

    def InitRHS(self, rhs):
        check(lib().iamrx_syncreg_init_rhs(self.h, rhs.h))

    def __del__(self):
        try:
            if self.h:
                lib().iamrx_syncreg_destroy(self.h)
                self.h = None
        except Exception:
            pass


def average_down(fine, crse, scomp=0, ncomp=None, ratio=2):
    check(lib().iamrx_average_down(fine.h, crse.h, scomp, crse.ncomp if ncomp is None else ncomp, ratio))


def abec_form(geom, coef, op, phi, rhs, out=None, rho=None, rho_comp=0, scale=1.0, bu=(1.0, 1.0, 1.0), beta=1.0, omega=1.15, lobc=(0, 0, 0), hibc=(0, 0, 0),
              maxorder=3):
    """one operation of the multigrid's finest-level kernel forms (include/iamrx.h: iamrx_abec_form)"""
    check(lib().iamrx_abec_form(C.byref(geom), int(coef), _h(rho), int(rho_comp), C.c_double(scale), (C.c_double * 3)(*[float(v) for v in bu]),
                                C.c_double(beta), int(op), phi.h, rhs.h, _h(out), C.c_double(omega), i3(lobc), i3(hibc), int(maxorder)))


def abec_gsrb_sweep(geom, alpha, beta, a, b, phi, rhs, omega=1.15, lobc=(0, 0, 0), hibc=(0, 0, 0), maxorder=2, fused=1):
    check(lib().iamrx_abec_gsrb_sweep(C.byref(geom), C.c_double(alpha), C.c_double(beta), _h(a), b[0].h, b[1].h, b[2].h,
                                      phi.h, rhs.h, C.c_double(omega), i3(lobc), i3(hibc), maxorder, int(fused)))


def abec_residual(geom, alpha, beta, a, b, out, phi, rhs=None, tensor=0):
    check(lib().iamrx_abec_residual(C.byref(geom), C.c_double(alpha), C.c_double(beta), _h(a), b[0].h, b[1].h, b[2].h,
                                    out.h, phi.h, _h(rhs), tensor))


def abec_solve(geom, alpha, beta, a, b, phi, rhs, lobc=(0, 0, 0), hibc=(0, 0, 0), rtol=1e-12, atol=1e-16, opts=None, tensor=0):
    st = MgStats()
    o = opts if opts is not None else mg_opts()
    check(lib().iamrx_abec_solve(C.byref(geom), C.c_double(alpha), C.c_double(beta), _h(a), b[0].h, b[1].h, b[2].h,
                                 phi.h, rhs.h, i3(lobc), i3(hibc), C.c_double(rtol), C.c_double(atol), C.byref(o), tensor, C.byref(st)))
    return st


def abec_solve_cf(geom, alpha, beta, a, b, phi, rhs, crse_phi, cgeom, ratio=2, lobc=(0, 0, 0), hibc=(0, 0, 0), rtol=1e-12, atol=1e-16, opts=None):
    """abec_solve on an AMR level that does not cover the domain: coarse/fine faces take Dirichlet data from crse_phi"""
    st = MgStats()
    o = opts if opts is not None else mg_opts()
    check(lib().iamrx_abec_solve_cf(C.byref(geom), C.c_double(alpha), C.c_double(beta), _h(a) if a is not None else None, _h(b[0]), _h(b[1]), _h(b[2]),
                                    _h(phi), _h(rhs), i3(lobc), i3(hibc), _h(crse_phi) if crse_phi is not None else None, C.byref(cgeom), ratio,
                                    C.c_double(rtol), C.c_double(atol), C.byref(o), C.byref(st)))
    return st


def mlmg_mac_solve(geom, umac, rho, rho_comp, S, mac_phi, rhs_scale, lobc=(0, 0, 0), hibc=(0, 0, 0),
                   mac_tol=1e-12, mac_abs_tol=1e-16, opts=None):
    """MacProj::mlmg_mac_solve (reference Source/MacProj.cpp:1084-1184)"""
    st = MgStats()
    o = opts if opts is not None else mg_opts(maxorder=4)
    check(lib().iamrx_mlmg_mac_solve(C.byref(geom), umac[0].h, umac[1].h, umac[2].h, rho.h, rho_comp, _h(S), mac_phi.h,
                                     C.c_double(rhs_scale), i3(lobc), i3(hibc), C.c_double(mac_tol), C.c_double(mac_abs_tol),
                                     C.byref(o), C.byref(st)))
    return st


def mlmg_mac_solve_cf(geom, umac, rho, rho_comp, S, mac_phi, rhs_scale, crse_phi, cgeom, ratio=2, lobc=(0, 0, 0), hibc=(0, 0, 0),
                      mac_tol=1e-12, mac_abs_tol=1e-16, opts=None):
    """MAC solve on an AMR level > 0: coarse/fine Dirichlet data from the coarse level's MAC phi (MacProj.cpp:1166-1170)"""
    st = MgStats()
    o = opts if opts is not None else mg_opts(maxorder=4)
    check(lib().iamrx_mlmg_mac_solve_cf(C.byref(geom), umac[0].h, umac[1].h, umac[2].h, rho.h, rho_comp, _h(S), mac_phi.h,
                                        C.c_double(rhs_scale), i3(lobc), i3(hibc), _h(crse_phi), C.byref(cgeom), ratio,
                                        C.c_double(mac_tol), C.c_double(mac_abs_tol), C.byref(o), C.byref(st)))
    return st


def derive_mag_vort(geom, out, vel, ocomp=0, vcomp=0):
    """|curl u| (dermgvort, NS_derive.cpp:86-264); vel with 1 filled ghost cell"""
    check(lib().iamrx_derive_mag_vort(C.byref(geom), _h(out), ocomp, _h(vel), vcomp))


TAG_GREATER, TAG_LESS, TAG_VORT, TAG_GRAD = 0, 1, 2, 3


def error_tag(geom, tags, field, mode, value, comp=0, level=0, realbox=None):
    """one amr.refinement_indicators entry (NS_error.cpp:10-145); tagged cells of `tags` are set to 1"""
    lo = hi = None
    if realbox is not None:
        lo = (C.c_double * 3)(*realbox[0]); hi = (C.c_double * 3)(*realbox[1])
    check(lib().iamrx_error_tag(C.byref(geom), _h(tags), _h(field), comp, mode, C.c_double(value), level, lo, hi))


def cluster_tags(geom, tags, blocking_factor=8, max_grid_size=32, grid_eff=0.7, n_error_buf=1, capacity=4096):
    """Berger-Rigoutsos grid generation from the tags of a level -> [(lo, hi), ...] in the index space of the tags"""
    buf = (C.c_int * (6 * capacity))()
    nb = C.c_int(capacity)
    check(lib().iamrx_cluster_tags(C.byref(geom), _h(tags), blocking_factor, max_grid_size, C.c_double(grid_eff), n_error_buf, buf, C.byref(nb)))
    return [(tuple(buf[6 * q:6 * q + 3]), tuple(buf[6 * q + 3:6 * q + 6])) for q in range(nb.value)]


def mac_sync_solve(geom, mac_reg, rho_half, dt, fine_layout, ucorr, mac_sync_phi, ratio=2, lobc=(0, 0, 0), hibc=(0, 0, 0),
                   tol=1e-10, abs_tol=1e-16, opts=None):
    """MacProj::mac_sync_solve (Source/MacProj.cpp:359-470)"""
    st = MgStats()
    o = opts if opts is not None else mg_opts(maxorder=4)
    check(lib().iamrx_mac_sync_solve(C.byref(geom), mac_reg.h, _h(rho_half), C.c_double(dt), fine_layout.h, ratio, _h(ucorr[0]), _h(ucorr[1]),
                                     _h(ucorr[2]), _h(mac_sync_phi), i3(lobc), i3(hibc), C.c_double(tol), C.c_double(abs_tol), C.byref(o),
                                     C.byref(st)))
    return st


def mac_divergence(geom, div, umac):
    check(lib().iamrx_mac_divergence(C.byref(geom), div.h, umac[0].h, umac[1].h, umac[2].h))


def _bcrec(ncomp, bc=None):
    arr = (C.c_int * (6 * ncomp))()
    if bc is not None:
        for n in range(ncomp):
            lo, hi = bc[n]
            for d in range(3):
                arr[6 * n + d] = int(lo[d])
                arr[6 * n + 3 + d] = int(hi[d])
    return arr


def godunov_extrap_vel_to_faces(geom, vel, force, umac, dt, bc=None, use_forces_in_trans=0, scheme=0):
    """Godunov::ExtrapVelToFaces (reference call site Source/NavierStokesBase.cpp:4487-4491)"""
    check(lib().iamrx_godunov_extrap_vel_to_faces(C.byref(geom), vel.h, _h(force), umac[0].h, umac[1].h, umac[2].h,
                                                  C.c_double(dt), _bcrec(3, bc), use_forces_in_trans, int(scheme)))


def godunov_compute_aofs(geom, aofs, acomp, S, ncomp, force, divu, umac, iconserv, dt, bc=None, is_velocity=0,
                         use_forces_in_trans=0, edge=None, flux=None, scheme=0):
    """kernel chain of NavierStokesBase::ComputeAofs (reference Source/NavierStokesBase.cpp:4594-4845)"""
    ic = (C.c_int * ncomp)(*[int(x) for x in iconserv])
    e = [None] * 3 if edge is None else edge
    f = [None] * 3 if flux is None else flux
    check(lib().iamrx_godunov_compute_aofs(C.byref(geom), aofs.h, acomp, S.h, ncomp, _h(force), _h(divu), umac[0].h, umac[1].h,
                                           umac[2].h, ic, C.c_double(dt), _bcrec(ncomp, bc), is_velocity, use_forces_in_trans,
                                           _h(e[0]), _h(e[1]), _h(e[2]), _h(f[0]), _h(f[1]), _h(f[2]), int(scheme)))


# ---- Diffusion operator entries on caller-owned data (include/iamrx.h; reference Source/Diffusion.H:53-225) ----------------------------
class DiffusionCrse(C.Structure):
    _fields_ = [("crse_old", C.c_void_p), ("crse_new", C.c_void_p), ("cgeom", C.c_void_p), ("ratio", C.c_int)]


def _mf3(v):
    return (C.c_void_p * 3)(*[m.h for m in v]) if v is not None else None


def _crse(crse):
    """crse = (crse_old or None, crse_new or None, cgeom, ratio) or None"""
    if crse is None:
        return None, None
    co, cn, cg, r = crse
    d = DiffusionCrse(_h(co) if co is not None else None, _h(cn) if cn is not None else None, C.cast(C.pointer(cg), C.c_void_p), int(r))
    return C.byref(d), d


def diffuse_scalar(geom, S_old, S_new, sigma, rho_comp, dt, theta, rho_half, rho_flag, betanp1, betan=None, fluxn=None, fluxnp1=None, delta_rhs=None,
                   rhs_comp=0, lobc=(0, 0, 0), hibc=(0, 0, 0), crse=None, add_old_time_divFlux=True, visc_tol=1e-10, opts=None, Rho_old=None, Rho_new=None):
    """Diffusion::diffuse_scalar (Source/Diffusion.cpp:207-599) on caller-owned MultiFabs"""
    st = MgStats()
    o = opts if opts is not None else mg_opts()
    cp, keep = _crse(crse)
    check(lib().iamrx_diffuse_scalar(C.byref(geom), _h(S_old), _h(Rho_old), S_new.h, _h(Rho_new), int(sigma), int(rho_comp), C.c_double(dt), C.c_double(theta),
                                     rho_half.h, int(rho_flag), _mf3(fluxn), _mf3(fluxnp1), _h(delta_rhs), int(rhs_comp), _mf3(betan), _mf3(betanp1),
                                     i3(lobc), i3(hibc), cp, int(bool(add_old_time_divFlux)), C.c_double(visc_tol), C.byref(o), C.byref(st)))
    return st


def diffuse_tensor_velocity(geom, U_old, U_new, rho_comp, dt, theta, rho_half, rho_flag, eta_np1, eta_n=None, visc_old_term=None, lobc=(0,) * 9, hibc=(0,) * 9,
                            crse=None, tflux=None, visc_tol=1e-10, opts=None, fill_new=None):
    """Diffusion::diffuse_tensor_velocity (Source/Diffusion.cpp:617-957); fill_new(): refill the ghost cells of U_new once it holds rho u*"""
    st = MgStats()
    o = opts if opts is not None else mg_opts()
    cp, keep = _crse(crse)
    CB = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p)
    cb = CB(lambda ctx, h: fill_new()) if fill_new is not None else C.cast(None, CB)
    check(lib().iamrx_diffuse_tensor_velocity(C.byref(geom), _h(U_old), U_new.h, int(rho_comp), C.c_double(dt), C.c_double(theta), rho_half.h, int(rho_flag),
                                              _h(visc_old_term), _mf3(eta_n), _mf3(eta_np1), (C.c_int * 9)(*lobc), (C.c_int * 9)(*hibc), cp, _mf3(tflux),
                                              C.c_double(visc_tol), C.byref(o), C.byref(st), cb, None))
    return st


def diffuse_tensor_vsync(geom, Vsync, dt, theta, rho_half, rho_flag, eta, bcrec_vel, Rho_old=None, Rho_new=None, rho_comp=3, lobc=(0,) * 9, hibc=(0,) * 9,
                         cgeom=None, ratio=2, tflux=None, visc_tol=1e-10, opts=None):
    """Diffusion::diffuse_tensor_Vsync (Source/Diffusion.cpp:1010-1178)"""
    st = MgStats()
    o = opts if opts is not None else mg_opts()
    check(lib().iamrx_diffuse_tensor_vsync(C.byref(geom), Vsync.h, C.c_double(dt), C.c_double(theta), rho_half.h, int(rho_flag), _h(Rho_old), _h(Rho_new), int(rho_comp),
                                           _mf3(eta), (C.c_int * 9)(*lobc), (C.c_int * 9)(*hibc), (C.c_int * 18)(*bcrec_vel),
                                           C.byref(cgeom) if cgeom is not None else None, int(ratio), _mf3(tflux), C.c_double(visc_tol), C.byref(o), C.byref(st)))
    return st


def diffuse_ssync(geom, Ssync, comp, dt, theta, rho_half, rho_flag, Rho_new, rho_comp, beta, lobc=(0, 0, 0), hibc=(0, 0, 0), cgeom=None, ratio=2, flux=None,
                  visc_tol=1e-10, opts=None):
    """Diffusion::diffuse_Ssync as NavierStokes::mac_sync calls it (Source/NavierStokes.cpp:1590-1640)"""
    st = MgStats()
    o = opts if opts is not None else mg_opts()
    check(lib().iamrx_diffuse_ssync(C.byref(geom), Ssync.h, int(comp), C.c_double(dt), C.c_double(theta), rho_half.h, int(rho_flag), Rho_new.h, int(rho_comp),
                                    _mf3(beta), i3(lobc), i3(hibc), C.byref(cgeom) if cgeom is not None else None, int(ratio), _mf3(flux), C.c_double(visc_tol),
                                    C.byref(o), C.byref(st)))
    return st


class ProjLevel(C.Structure):
    """iamrx_proj_level: a level of a multi-level nodal projection on caller-owned data"""
    _fields_ = [("geom", C.c_void_p), ("layout", C.c_void_p), ("lobc", C.c_int * 3), ("hibc", C.c_int * 3), ("ratio", C.c_int), ("gp", C.c_void_p)]


def mlsync_project(crse, fine, pres_crse, vel_crse, pres_fine, vel_fine, rho_crse, rho_fine, Vsync, V_corr, phi_crse, phi_fine, rhs_sync_reg, dt,
                   crse_sync_reg=None, crse_iteration=1, crse_dt_ratio=1, vcomp_crse=0, vcomp_fine=0, sync_tol=1e-10, abs_tol=1e-16, opts=None):
    """Projection::MLsyncProject (Source/Projection.cpp:457-607) on caller-owned arrays; crse / fine: (geom, layout, lobc, hibc, ratio, gp or None)"""
    st = MgStats()
    o = opts if opts is not None else mg_opts()
    keep = []

    def mk(lv):
        g, lay, lo, hi, r, gp = lv
        keep.append(g)
        return ProjLevel(C.cast(C.pointer(g), C.c_void_p), lay.h, i3(lo), i3(hi), int(r), _h(gp) if gp is not None else None)
    pc, pf = mk(crse), mk(fine)
    check(lib().iamrx_mlsync_project(C.byref(pc), C.byref(pf), pres_crse.h, vel_crse.h, int(vcomp_crse), pres_fine.h, vel_fine.h, int(vcomp_fine), rho_crse.h,
                                     rho_fine.h, Vsync.h, V_corr.h, phi_crse.h, phi_fine.h, rhs_sync_reg.h, crse_sync_reg.h if crse_sync_reg is not None else None,
                                     C.c_double(dt), int(crse_iteration), int(crse_dt_ratio), C.c_double(sync_tol), C.c_double(abs_tol), C.byref(o), C.byref(st)))
    return st


# ---- round 6: MacProj::mac_sync_compute, SyncRegister::CompAdd, Projection::initial*Project on caller-owned data ----------------------
def mac_sync_compute(geom, ucorr, Vsync, Ssync, S_vel, S_scal, nscal, gradp, umac, iconserv_scal, dt, visc_vel=None, tforce_scal=None, divu=None,
                     do_mom_diff=0, gravity=0.0, bc_vel=None, bc_scal=None, use_forces_in_trans=0, scheme=0, flux_vel=None, flux_scal=None):
    """MacProj::mac_sync_compute as NavierStokes::mac_sync calls it (Source/MacProj.cpp:488-731)"""
    ic = (C.c_int * nscal)(*[int(x) for x in iconserv_scal])
    fv = [None] * 3 if flux_vel is None else flux_vel
    fs = [None] * 3 if flux_scal is None else flux_scal
    check(lib().iamrx_mac_sync_compute(C.byref(geom), ucorr[0].h, ucorr[1].h, ucorr[2].h, Vsync.h, Ssync.h, S_vel.h, S_scal.h, int(nscal), _h(visc_vel),
                                       _h(tforce_scal), gradp.h, _h(divu), umac[0].h, umac[1].h, umac[2].h, ic, int(do_mom_diff), C.c_double(gravity),
                                       C.c_double(dt), _bcrec(3, bc_vel), _bcrec(nscal, bc_scal), int(use_forces_in_trans), int(scheme),
                                       _h(fv[0]), _h(fv[1]), _h(fv[2]), _h(fs[0]), _h(fs[1]), _h(fs[2])))


def mac_sync_compute_edge(geom, ucorr, Sync, sync_indx, edge, edge_comp, flux=None):
    """MacProj::mac_sync_compute with known edge states (Source/MacProj.cpp:733-786)"""
    f = [None] * 3 if flux is None else flux
    check(lib().iamrx_mac_sync_compute_edge(C.byref(geom), ucorr[0].h, ucorr[1].h, ucorr[2].h, Sync.h, int(sync_indx), edge[0].h, edge[1].h, edge[2].h,
                                            int(edge_comp), _h(f[0]), _h(f[1]), _h(f[2])))


def _proj_levels(levels, keep):
    arr = (ProjLevel * len(levels))()
    for q, (g, lay, lo, hi, r, gp) in enumerate(levels):
        keep.append(g)
        arr[q] = ProjLevel(C.cast(C.pointer(g), C.c_void_p), lay.h, i3(lo), i3(hi), int(r), _h(gp) if gp is not None else None)
    return arr


def _hv(mfs, n):
    a = (C.c_void_p * n)()
    for q in range(n):
        a[q] = None if (mfs is None or mfs[q] is None) else mfs[q].h
    return a


def initial_velocity_project(levels, vel, vcomp, pres, rho=None, rho_comp=None, divu=None, divu_comp=None, proj_tol=1e-12, proj_abs_tol=1e-16, opts=None):
    """Projection::initialVelocityProject (Source/Projection.cpp:615-838); levels: [(geom, layout, lobc, hibc, ratio, gp or None)], coarsest first"""
    n = len(levels)
    st = MgStats(); o = opts if opts is not None else mg_opts(); keep = []
    pl = _proj_levels(levels, keep)
    vc = (C.c_int * n)(*[int(v) for v in vcomp])
    rc = (C.c_int * n)(*[int(v) for v in (rho_comp or [0] * n)])
    dc = (C.c_int * n)(*[int(v) for v in (divu_comp or [0] * n)])
    check(lib().iamrx_initial_velocity_project(n, pl, _hv(vel, n), vc, _hv(pres, n), _hv(rho, n) if rho is not None else None, rc,
                                               _hv(divu, n) if divu is not None else None, dc, C.c_double(proj_tol), C.c_double(proj_abs_tol), C.byref(o), C.byref(st)))
    return st


def initial_sync_project(levels, vel_new, vcomp, vel_old, phi, rho_half, dt, pres_new=None, divu_new=None, divu_old=None, divu_comp=None, proj_tol=1e-12,
                         proj_abs_tol=1e-16, opts=None):
    """Projection::initialSyncProject (Source/Projection.cpp:970-1185)"""
    n = len(levels)
    st = MgStats(); o = opts if opts is not None else mg_opts(); keep = []
    pl = _proj_levels(levels, keep)
    vc = (C.c_int * n)(*[int(v) for v in vcomp])
    dc = (C.c_int * n)(*[int(v) for v in (divu_comp or [0] * n)])
    check(lib().iamrx_initial_sync_project(n, pl, _hv(vel_new, n), vc, _hv(vel_old, n), _hv(phi, n), _hv(pres_new, n) if pres_new is not None else None,
                                           _hv(rho_half, n), _hv(divu_new, n) if divu_new is not None else None,
                                           _hv(divu_old, n) if divu_old is not None else None, dc, C.c_double(dt), C.c_double(proj_tol),
                                           C.c_double(proj_abs_tol), C.byref(o), C.byref(st)))
    return st
