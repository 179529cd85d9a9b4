"""Hierarchy of NavierStokes levels (ctypes view of iamrx_amr_*; include/iamrx.h): Amr::coarseTimeStep with subcycling,
NavierStokesBase::post_timestep (reflux, avgDown, mac_sync, level_sync) and the multi-level NavierStokes::post_init."""
import ctypes as C
from .lib import lib, check, mg_opts, MgStats
from .ns import NavierStokes, ns_params


class TagRule(C.Structure):
    _fields_ = [("comp", C.c_int), ("mode", C.c_int), ("nvalue", C.c_int), ("max_level", C.c_int), ("has_box", C.c_int),
                ("value", C.c_double * 8), ("box_lo", C.c_double * 3), ("box_hi", C.c_double * 3)]


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
        self._refresh(keep_layouts=True)

    def _refresh(self, keep_layouts=False):
        """(re)build the Python views of the levels; after a regrid the layouts are re-created from the hierarchy's box lists"""
        from .lib import Layout
        n = C.c_int()
        check(lib().iamrx_amr_nlevels(self.h, C.byref(n)))
        if not keep_layouts:
            lays = [self.layouts[0]]
            for l in range(1, n.value):
                nb = C.c_int(0)
                check(lib().iamrx_amr_level_boxes(self.h, l, C.byref(nb), None))
                arr = (C.c_int * (6 * nb.value))()
                check(lib().iamrx_amr_level_boxes(self.h, l, C.byref(nb), arr))
                hl = C.c_void_p()
                check(lib().iamrx_amr_level_layout(self.h, l, C.byref(hl)))
                lays.append(Layout.from_handle(hl, [(tuple(arr[6 * q:6 * q + 3]), tuple(arr[6 * q + 3:6 * q + 6])) for q in range(nb.value)]))
            self.layouts = lays
        self.levels = []
        for l in range(n.value):
            hl = C.c_void_p()
            check(lib().iamrx_amr_level(self.h, l, C.byref(hl)))
            self.levels.append(_Level(hl, self.level_geom(l), self.layouts[l], self.params, self.opts))

    def set_regrid(self, max_level, regrid_int, rules, blocking_factor=8, max_grid_size=32, grid_eff=0.7, n_error_buf=1, compute_new_dt_on_regrid=0,
                   do_refine_outflow=0, do_derefine_outflow=1, nbuf_outflow=1):
        """rules: list of dicts(comp (0..4, -1 = mag_vort), mode (0 greater, 1 less, 2 vorticity, 3 adjacent difference), value (list per level),
        max_level (optional), box_lo / box_hi (optional)) -- amr.refinement_indicators of NS_error.cpp"""
        arr = (TagRule * max(1, len(rules)))()
        for q, r in enumerate(rules):
            arr[q].comp, arr[q].mode = int(r.get("comp", 4)), int(r.get("mode", 0))
            vals = list(r["value"]) if hasattr(r["value"], "__len__") else [r["value"]]
            arr[q].nvalue = len(vals)
            for i, v in enumerate(vals[:8]):
                arr[q].value[i] = float(v)
            arr[q].max_level = int(r.get("max_level", 1000))
            arr[q].has_box = 1 if "box_lo" in r else 0
            for d in range(3):
                arr[q].box_lo[d] = float(r.get("box_lo", (0, 0, 0))[d])
                arr[q].box_hi[d] = float(r.get("box_hi", (0, 0, 0))[d])
        check(lib().iamrx_amr_set_regrid(self.h, int(max_level), int(regrid_int), int(blocking_factor), int(max_grid_size), C.c_double(grid_eff),
                                         int(n_error_buf), len(rules), arr))
        check(lib().iamrx_amr_set_compute_new_dt_on_regrid(self.h, int(compute_new_dt_on_regrid)))
        check(lib().iamrx_amr_set_outflow_tagging(self.h, int(do_refine_outflow), int(do_derefine_outflow), int(nbuf_outflow)))

    def regrid_log(self):
        """[(lbase, time, [boxes of level lbase + 1, boxes of level lbase + 2, ...]), ...] of the last coarse step"""
        n = C.c_int()
        check(lib().iamrx_amr_regrid_log_count(self.h, C.byref(n)))
        out = []
        for e in range(n.value):
            lb, tm, nl = C.c_int(), C.c_double(), C.c_int()
            nb = (C.c_int * 8)()
            check(lib().iamrx_amr_regrid_log_event(self.h, e, C.byref(lb), C.byref(tm), C.byref(nl), nb, None))
            tot = sum(nb[q] for q in range(nl.value))
            arr = (C.c_int * (6 * max(tot, 1)))()
            check(lib().iamrx_amr_regrid_log_event(self.h, e, C.byref(lb), C.byref(tm), C.byref(nl), nb, arr))
            grids, q = [], 0
            for l in range(nl.value):
                grids.append([(tuple(arr[6 * (q + b):6 * (q + b) + 3]), tuple(arr[6 * (q + b) + 3:6 * (q + b) + 6])) for b in range(nb[l])])
                q += nb[l]
            out.append((lb.value, tm.value, grids))
        return out

    def regrid(self):
        ch = C.c_int()
        check(lib().iamrx_amr_regrid(self.h, C.byref(ch)))
        if ch.value:
            self._refresh()
        return bool(ch.value)

    def install_grids(self, grids):
        """grids: per refined level the list of (lo, hi) boxes in that level's index space"""
        nb = (C.c_int * max(1, len(grids)))(*[len(g) for g in grids])
        flat = [v for g in grids for lo, hi in g for v in (*lo, *hi)]
        arr = (C.c_int * max(1, len(flat)))(*flat)
        ch = C.c_int()
        check(lib().iamrx_amr_install_grids(self.h, len(grids), nb, arr, C.byref(ch)))
        if ch.value:
            self._refresh()
        return bool(ch.value)

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
        n = C.c_int()
        check(lib().iamrx_amr_nlevels(self.h, C.byref(n)))
        boxes_now = []
        for l in range(1, n.value):
            nb = C.c_int(0)
            check(lib().iamrx_amr_level_boxes(self.h, l, C.byref(nb), None))
            boxes_now.append(nb.value)
        if n.value != len(self.levels) or self._grids_changed():
            self._refresh()
        return dt.value

    def _grids_changed(self):
        for l in range(1, len(self.levels)):
            nb = C.c_int(0)
            check(lib().iamrx_amr_level_boxes(self.h, l, C.byref(nb), None))
            arr = (C.c_int * (6 * max(1, nb.value)))()
            check(lib().iamrx_amr_level_boxes(self.h, l, C.byref(nb), arr))
            now = [(tuple(arr[6 * q:6 * q + 3]), tuple(arr[6 * q + 3:6 * q + 6])) for q in range(nb.value)]
            if now != self.layouts[l].boxes:
                return True
        return False

    @property
    def time(self):
        t = C.c_double()
        check(lib().iamrx_amr_time(self.h, C.byref(t), None))
        return t.value

    def dts(self):
        arr = (C.c_double * self.nlev)()
        check(lib().iamrx_amr_time(self.h, None, arr))
        return list(arr)

    def profile(self, enable):
        """-> (hierarchy sections [16], per-level sections [nlev][8]) accumulated so far; then enable: 1 reset + start, 0 stop, -1 keep"""
        sec = (C.c_double * 16)()
        lv = (C.c_double * (8 * self.nlev))()
        check(lib().iamrx_amr_profile(self.h, int(enable), sec, lv))
        return list(sec), [list(lv[8 * l:8 * l + 8]) for l in range(self.nlev)]

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
