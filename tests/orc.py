"""ctypes binding of the CPU oracle (oracle/liborc.so) for the test-suite.

Test infrastructure only: nothing under iamr_amd/ imports this module.
Arrays use the AMReX Array4 layout (i fastest, component slowest): a numpy array of shape
(nx, ny, nz, nc) in Fortran order.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
_LIB = None


def build():
    subprocess.run(["make", "-s", "-C", ORC_DIR], check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ORC_DIR, "liborc.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _declare(_LIB)
    return _LIB


class CFab(C.Structure):
    _fields_ = [("p", C.POINTER(C.c_double)), ("lo", C.c_int * 3), ("hi", C.c_int * 3), ("nc", C.c_int)]


class CGeom(C.Structure):
    _fields_ = [("n", C.c_int * 3), ("dx", C.c_double * 3), ("problo", C.c_double * 3), ("periodic", C.c_int * 3)]


class CBCRec(C.Structure):
    _fields_ = [("lo", C.c_int * 3), ("hi", C.c_int * 3)]


class CAbecLevel(C.Structure):
    _fields_ = [("g", CGeom), ("alpha", C.c_double), ("beta", C.c_double), ("a", CFab), ("b", CFab * 3),
                ("ncomp", C.c_int), ("tensor", C.c_int), ("bc_percomp", C.c_int),
                ("nbox", C.c_int), ("boxes", C.POINTER(C.c_int)), ("cf_loc", C.c_double * 3)]


class CMgStats(C.Structure):
    _fields_ = [("iters", C.c_int), ("resnorm0", C.c_double), ("rhsnorm0", C.c_double), ("resnorm", C.c_double),
                ("bottom_iters_total", C.c_int), ("converged", C.c_int)]


class CMgOpts(C.Structure):
    _fields_ = [("nu1", C.c_int), ("nu2", C.c_int), ("nuf", C.c_int), ("nub", C.c_int), ("max_iters", C.c_int),
                ("bottom_maxiter", C.c_int), ("bottom_reltol", C.c_double), ("omega", C.c_double),
                ("maxorder", C.c_int), ("max_coarsening_level", C.c_int), ("min_width", C.c_int),
                ("nodal_sweeps", C.c_int), ("nodal_smoother", C.c_int), ("verbose", C.c_int),
                ("bottom_smoother_only", C.c_int), ("fixed_iters", C.c_int)]


class CNsParams(C.Structure):
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


PF = C.POINTER(CFab)


def _declare(L):
    L.orc_slope4.restype = C.c_double
    if hasattr(L, "orc_ns_create"):
        L.orc_ns_create.restype = C.c_void_p
        L.orc_ns_fab.restype = PF
        L.orc_ns_step.restype = C.c_double
        L.orc_ns_time.restype = C.c_double
        L.orc_ns_dt.restype = C.c_double


CELL = (0, 0, 0)
NODE = (1, 1, 1)


def face(d):
    t = [0, 0, 0]
    t[d] = 1
    return tuple(t)


class Fab:
    """numpy-backed array on cells [0,n-1] converted to `typ` and grown by ng."""

    def __init__(self, n, typ=CELL, ng=0, nc=1, fill=0.0, lo=None, hi=None):
        if lo is None:
            lo = [-ng] * 3
            hi = [n[d] - 1 + typ[d] + ng for d in range(3)]
        self.lo = list(lo)
        self.hi = list(hi)
        self.nc = nc
        shape = tuple(self.hi[d] - self.lo[d] + 1 for d in range(3)) + (nc,)
        self.a = np.full(shape, fill, dtype=np.float64, order="F")

    @property
    def c(self):
        f = CFab()
        f.p = self.a.ctypes.data_as(C.POINTER(C.c_double))
        f.lo = (C.c_int * 3)(*self.lo)
        f.hi = (C.c_int * 3)(*self.hi)
        f.nc = self.nc
        self._c = f
        return f

    def ref(self):
        return C.byref(self.c)

    def valid(self, n, typ=CELL):
        """view of the valid region (cells [0,n-1] converted to typ)"""
        sl = tuple(slice(-self.lo[d], -self.lo[d] + n[d] + typ[d]) for d in range(3))
        return self.a[sl]

    def copy(self):
        g = Fab.__new__(Fab)
        g.lo, g.hi, g.nc = list(self.lo), list(self.hi), self.nc
        g.a = self.a.copy(order="F")
        return g


def from_cfab(pf):
    """numpy view (no copy) of a CFab owned by the oracle"""
    f = pf.contents
    shape = tuple(f.hi[d] - f.lo[d] + 1 for d in range(3)) + (f.nc,)
    size = int(np.prod(shape))
    arr = np.ctypeslib.as_array(f.p, shape=(size,)).reshape(shape, order="F")
    out = Fab.__new__(Fab)
    out.lo, out.hi, out.nc = list(f.lo), list(f.hi), f.nc
    out.a = arr
    return out


def geom(n, problo=(0.0, 0.0, 0.0), probhi=(1.0, 1.0, 1.0), periodic=(1, 1, 1)):
    g = CGeom()
    g.n = (C.c_int * 3)(*n)
    g.dx = (C.c_double * 3)(*[(probhi[d] - problo[d]) / n[d] for d in range(3)])
    g.problo = (C.c_double * 3)(*problo)
    g.periodic = (C.c_int * 3)(*periodic)
    return g


# amrex::MLNodeLaplacian / MLMG's V-cycle shape (4 Gauss-Seidel sweeps per smooth call, nu1 = nu2 = 2), which the oracle restates.  The
# product's default is a shorter cycle (iamrx_mg_opts: 2 sweeps, 1 + 1 calls; same converged solution); tests that compare iterates or
# iteration counts with the oracle ask the product for the upstream shape.
# upstream's V-cycle shape and hierarchy (coarsened to 2^3, host-driven BiCGStab there): what the oracle runs -> equal iteration counts
UPSTREAM_NODAL_CYCLE = dict(nodal_sweeps=4, nodal_nu1=2, nodal_nu2=2, device_bottom=0)


def mg_opts(**kw):
    o = CMgOpts()
    lib().orc_mg_default_opts(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def bcrecs(ncomp, lo=(0, 0, 0), hi=(0, 0, 0)):
    arr = (CBCRec * ncomp)()
    for n in range(ncomp):
        arr[n].lo = (C.c_int * 3)(*lo)
        arr[n].hi = (C.c_int * 3)(*hi)
    return arr


def i3(v):
    return (C.c_int * 3)(*v)


def fabptrs(fabs):
    arr = (PF * 3)()
    keep = []
    for d in range(3):
        cf = fabs[d].c
        keep.append(cf)
        arr[d] = C.pointer(cf)
    arr._keep = keep
    return arr


def abec_level(g, b, alpha=0.0, beta=1.0, a=None, ncomp=1, tensor=0, bc_percomp=0, boxes=None, ratio=2):
    """boxes: [(lo, hi), ...] of an AMR level that does not cover the domain (coarse/fine faces, see orc.h)"""
    L = CAbecLevel()
    if boxes:
        flat = [v for lo, hi in boxes for v in (*lo, *hi)]
        L._boxes_keep = (C.c_int * len(flat))(*flat)
        L.nbox = len(boxes)
        L.boxes = C.cast(L._boxes_keep, C.POINTER(C.c_int))
        for d in range(3):
            L.cf_loc[d] = 0.5 * ratio * g.dx[d]
    L.g = g
    L.alpha = alpha
    L.beta = beta
    if a is not None:
        L.a = a.c
    for d in range(3):
        L.b[d] = b[d].c
    L.ncomp = ncomp
    L.tensor = tensor
    L.bc_percomp = bc_percomp
    return L


# ---- multi-level hierarchy (oracle/orc_amr.c) ---------------------------------------------------------------------------
def ns_params(**kw):
    p = CNsParams()
    lib().orc_ns_default_params(C.byref(p))
    for k, v in kw.items():
        if isinstance(v, (list, tuple)):
            arr = getattr(p, k)
            for q, x in enumerate(v):
                arr[q] = x
        else:
            setattr(p, k, v)
    return p


class OrcAmr:
    """levels: list of box lists; levels[0] is ignored (the base level covers the domain), levels[l] = [(lo, hi), ...] in the
    index space of level l."""

    def __init__(self, g0, params, opts, levels, ratio=2):
        L = lib()
        L.orc_amr_create.restype = C.c_void_p
        L.orc_amr_level.restype = C.c_void_p
        L.orc_amr_cov.restype = PF
        L.orc_amr_coarse_step.restype = C.c_double
        L.orc_amr_time.restype = C.c_double
        L.orc_amr_dt.restype = C.c_double
        self.nlev = len(levels)
        self.g0 = g0
        self.ratio = ratio
        nbox = (C.c_int * self.nlev)(*[len(b) for b in levels])
        flat = [v for lev in levels[1:] for lo, hi in lev for v in (*lo, *hi)]
        boxes = (C.c_int * max(1, len(flat)))(*flat)
        self.h = C.c_void_p(L.orc_amr_create(C.byref(g0), C.byref(params), C.byref(opts), self.nlev, ratio, nbox, boxes))
        assert self.h
        self.levels = levels

    def n(self, lev):
        return [self.g0.n[d] * self.ratio ** lev for d in range(3)]

    def dx(self, lev):
        return [self.g0.dx[d] / self.ratio ** lev for d in range(3)]

    def fab(self, lev, which):
        L = lib()
        ns = C.c_void_p(L.orc_amr_level(self.h, lev))
        return from_cfab(L.orc_ns_fab(ns, which))

    def cov(self, lev):
        if lev == 0:
            return np.ones(tuple(self.n(0)), dtype=bool)
        return from_cfab(lib().orc_amr_cov(self.h, lev)).a[..., 0] != 0.0

    def post_init(self, stop_time=-1.0):
        lib().orc_amr_post_init(self.h, C.c_double(stop_time))

    def step(self):
        return lib().orc_amr_coarse_step(self.h)

    def time(self):
        return lib().orc_amr_time(self.h)

    def regrid_then_step(self, grids, compute_new_dt_on_regrid=0):
        """the coarse step during which the hierarchy is regridded to `grids` (per refined level a list of (lo, hi) in that level's
        index space): computeNewDt, Amr::regrid with these grids, computeNewDt(post_regrid_flag = 1) if amr.compute_new_dt_on_regrid, timeStep"""
        L = lib()
        L.orc_amr_coarse_step_post_regrid.restype = C.c_double
        L.orc_amr_compute_new_dt(self.h)
        nb = (C.c_int * max(1, len(grids)))(*[len(g) for g in grids])
        flat = [v for g in grids for lo, hi in g for v in (*lo, *hi)]
        arr = (C.c_int * max(1, len(flat)))(*flat)
        L.orc_amr_regrid(self.h, len(grids), nb, arr)
        self.nlev = len(grids) + 1
        self.levels = [[]] + [list(g) for g in grids]
        return L.orc_amr_coarse_step_post_regrid(self.h, C.c_int(compute_new_dt_on_regrid))

    def step_with_regrids(self, events, compute_new_dt_on_regrid=0):
        """one coarse step that replays the regrids the product did during ITS coarse step (iamr_amd.amr.Amr.regrid_log): events =
        [(lbase, time, [boxes of level lbase + 1, ...]), ...].  A level-0 event is the regrid at the start of the step; events with
        lbase > 0 (Amr::timeStep's okToRegrid(i) for i >= 1) are scheduled and applied when the oracle's subcycling reaches their level
        and time."""
        L = lib()
        L.orc_amr_coarse_step_post_regrid.restype = C.c_double
        L.orc_amr_compute_new_dt(self.h)
        L.orc_amr_clear_regrid_schedule()
        post = 0
        cur = [list(g) for g in self.levels]
        for lbase, tm, grids in events:
            nb = (C.c_int * max(1, len(grids)))(*[len(g) for g in grids])
            flat = [v for g in grids for lo, hi in g for v in (*lo, *hi)]
            arr = (C.c_int * max(1, len(flat)))(*flat)
            if lbase == 0:
                L.orc_amr_regrid(self.h, len(grids), nb, arr)
                post = compute_new_dt_on_regrid
            else:
                L.orc_amr_schedule_regrid(C.c_int(lbase), C.c_double(tm), len(grids), nb, arr)
            cur = cur[:lbase + 1] + [list(g) for g in grids]
        dt = L.orc_amr_coarse_step_post_regrid(self.h, C.c_int(post))
        L.orc_amr_clear_regrid_schedule()
        self.levels = cur
        self.nlev = len(cur)
        return dt

    def dt(self, lev):
        return lib().orc_amr_dt(self.h, C.c_int(lev))

    def sync_stats(self):
        st = CMgStats()
        lib().orc_amr_sync_stats(self.h, C.byref(st))
        return st

    def cell_centres(self, lev):
        n, dx = self.n(lev), self.dx(lev)
        ax = [self.g0.problo[d] + (np.arange(n[d]) + 0.5) * dx[d] for d in range(3)]
        return np.meshgrid(*ax, indexing="ij")

    def set_state(self, lev, arr):
        """arr: (nx, ny, nz, 5) valid data of the level's whole index space"""
        f = self.fab(lev, 0)
        n = self.n(lev)
        f.valid(n)[...] = arr

    def state(self, lev, which=0):
        return self.fab(lev, which).valid(self.n(lev)).copy()

    def __del__(self):
        try:
            lib().orc_amr_destroy(self.h)
        except Exception:
            pass


def taylorgreen_state(X, Y, Z, vfac=1.0, a=1.0, b=1.0, c=1.0, rho0=1.0):
    tp = 2.0 * np.pi
    S = np.zeros(X.shape + (5,), order="F")
    S[..., 0] = vfac * np.sin(a * tp * X) * np.cos(b * tp * Y) * np.cos(c * tp * Z)
    S[..., 1] = -vfac * np.cos(a * tp * X) * np.sin(b * tp * Y) * np.cos(c * tp * Z)
    S[..., 3] = rho0
    S[..., 4] = (rho0 * vfac * vfac / 16.0) * (2.0 + np.cos(2.0 * c * tp * Z)) * (np.cos(2.0 * a * tp * X) + np.cos(2.0 * b * tp * Y))
    return S
