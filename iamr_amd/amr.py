"""Hierarchy of NavierStokes levels (ctypes view of iamrx_amr_*; include/iamrx.h): Amr::coarseTimeStep with subcycling,
NavierStokesBase::post_timestep (reflux, avgDown, mac_sync, level_sync) and the multi-level NavierStokes::post_init."""
import ctypes as C
from .lib import lib, check, mg_opts, MgStats
from .ns import NavierStokes, ns_params


class _Level(NavierStokes):
    """a level borrowed from an Amr hierarchy (never destroyed on its own)"""

    def __init__(self, handle, geom, layout, params, opts):
        self.geom = geom
        self.layout = layout
        self.params = params
        self.opts = opts
        self.h = handle

    def __del__(self):
        pass


class Amr:
    def __init__(self, geom0, layouts, params=None, opts=None, ratio=2):
        self.geom0 = geom0
        self.layouts = list(layouts)
        self.ratio = ratio
        self.params = params if params is not None else ns_params()
        self.opts = opts if opts is not None else mg_opts()
        self.h = C.c_void_p()
        arr = (C.c_void_p * len(self.layouts))(*[l.h for l in self.layouts])
        check(lib().iamrx_amr_create(C.byref(geom0), len(self.layouts), arr, int(ratio), C.byref(self.params), C.byref(self.opts), C.byref(self.h)))
        self.levels = []
        for l in range(len(self.layouts)):
            hl = C.c_void_p()
            check(lib().iamrx_amr_level(self.h, l, C.byref(hl)))
            self.levels.append(_Level(hl, self.level_geom(l), self.layouts[l], self.params, self.opts))

    def level_geom(self, l):
        from .lib import Geom
        n = [self.geom0.n[d] * self.ratio ** l for d in range(3)]
        return Geom.make(n, tuple(self.geom0.prob_lo), tuple(self.geom0.prob_hi), tuple(self.geom0.periodic))

    @property
    def nlev(self):
        return len(self.layouts)

    def post_init(self, stop_time=-1.0):
        check(lib().iamrx_amr_post_init(self.h, C.c_double(stop_time)))

    def coarse_step(self):
        dt = C.c_double()
        check(lib().iamrx_amr_coarse_step(self.h, C.byref(dt)))
        return dt.value

    @property
    def time(self):
        t = C.c_double()
        check(lib().iamrx_amr_time(self.h, C.byref(t), None))
        return t.value

    def dts(self):
        arr = (C.c_double * self.nlev)()
        check(lib().iamrx_amr_time(self.h, None, arr))
        return list(arr)

    def sync_stats(self):
        a, b = MgStats(), MgStats()
        check(lib().iamrx_amr_sync_stats(self.h, C.byref(a), C.byref(b)))
        return a, b

    def __del__(self):
        try:
            if self.h:
                self.levels = []
                lib().iamrx_amr_destroy(self.h)
        except Exception:
            pass
