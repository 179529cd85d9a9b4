"""ParmParse-compatible inputs for the level driver.

Reads IAMR `inputs.*` / `regtest.*` files (`key = value ...`, `#` comments, later definitions override earlier ones, trailing
`key=value` command-line arguments override the file -- the amrex::ParmParse rules used by Source/main.cpp:60-75) and maps the
keys this library implements onto `ns_params`, the geometry and the box layout:

  amr.n_cell, amr.max_level (must be 0), amr.max_grid_size (32), geometry.prob_lo / prob_hi / is_periodic / coord_sys (0),
  ns.cfl, ns.init_iter, ns.init_vel_iter, ns.init_shrink, ns.change_max, ns.fixed_dt, ns.init_dt, ns.gravity,
  ns.be_cn_theta, ns.do_mom_diff, ns.do_cons_trac, ns.vel_visc_coef, ns.scal_diff_coefs, ns.lo_bc, ns.hi_bc, ns.advection_scheme,
  ns.visc_tol, godunov.use_forces_in_trans, mac_proj.mac_tol / mac_abs_tol, proj.proj_tol / proj_abs_tol,
  {x,y,z}{lo,hi}.velocity / .density / .tracer, prob.probtype (1: fluid at rest, 4: constant velocity + blob, 5: DoubleShearLayer, 7: Euler, 10: RayleighTaylor, 11: TaylorGreen), prob.velocity_factor, prob.a/b/c,
  prob.density_ic, prob.rho_1 / rho_2 / tra_1 / tra_2 / interface_width / perturbation_amplitude (probtype 10), max_step, stop_time
(reference: Source/NavierStokesBase.cpp:431-557, Source/NavierStokes.cpp:250-310, Source/MacProj.cpp:62-75,
Source/Projection.cpp:49-65, Source/Diffusion.cpp:98-118, Source/prob/prob_init.cpp:8-60, Source/main.cpp:60-145).
Keys that select features this library does not have (AMR levels, EB, particles, inflow/outflow ...) raise; keys that only
concern I/O or verbosity are ignored and listed in `Inputs.ignored`.  Host-only code: no GPU needed to parse."""
import os
import re

# I/O, verbosity and grid-generation keys that do not change the numbers of a fixed-grid run: matched EXACTLY or, for the entries
# ending in ".", as a prefix of a whole ParmParse namespace
_IGNORED_KEYS = ("amr.v", "amr.verbose", "ns.v", "ns.verbose", "proj.v", "proj.verbose", "mac_proj.v", "mac_proj.verbose", "mac.v", "diffuse.v",
                 "diffuse.verbose", "nodal_proj.verbose", "ns.sum_interval", "ns.getForceVerbose", "amr.grid_log", "amr.probin_file",
                 "amr.blocking_factor", "amr.regrid_int", "amr.ref_ratio", "amr.regrid_file", "amr.initial_grid_file", "amr.refinement_indicators", "amr.n_error_buf", "amr.grid_eff",
                 "amr.check_per", "amr.checkpoint_files_output", "amr.plot_files_output", "amr.plot_per",
                 "amr.plotfile_on_restart", "amr.checkpoint_on_restart")
_IGNORED_NAMESPACES = ("mg.", "fab.", "fabarray.", "amrex.", "amr.refinement_indicators")
# boundary values of the second tracer: read by upstream only with ns.do_trac2 = 1 (which raises here), unused otherwise
# keys that switch physics or start-up paths this library does not have: their reference defaults are accepted, anything else raises
_UNIMPLEMENTED_UNLESS = {"ns.variable_vel_visc": "0", "ns.variable_scal_diff": "0", "ns.do_init_proj": "1", "ns.do_mac_proj": "1",
                         "ns.do_init_vort_proj": "0", "ns.do_divu_sync": "0", "ns.do_scalar_update_in_order": "0",
                         # the multi-level step always refluxes, sync-projects and subcycles (NavierStokesBase.cpp:461-462, 2562-2582;
                         # Amr's subcycling_mode "Auto" with ratio 2 = one fine step per direction of refinement)
                         "ns.do_reflux": "1", "ns.do_sync_proj": "1", "amr.subcycling_mode": "Auto"}


def read_grid_file(path, ref_ratio, slab=None):
    """amr.regrid_file / amr.initial_grid_file (upstream Amr::readProbinFile-era format, e.g. Exec/run2d/test_grids/fixed_grids_2): number
    of refined levels, then per level the number of boxes and the boxes `((lo) (hi) (type))` given in the index space of the NEXT
    COARSER level.  Returns, per refined level, the boxes refined by ref_ratio[level-1] -- the level's own index space (pinned on the
    reference's committed plotfile of the same run, tests/test_cpu_plotfile.py)."""
    toks = open(path).read().split("\n")
    lines = [t.split("#", 1)[0].strip() for t in toks]
    lines = [t for t in lines if t]
    nlev = int(lines[0])
    out, q = [], 1
    for l in range(nlev):
        nb = int(lines[q]); q += 1
        boxes = []
        for _ in range(nb):
            m = re.match(r"\(\(([-\d, ]+)\)\s*\(([-\d, ]+)\)\s*\(([-\d, ]+)\)\)", lines[q]); q += 1
            if not m:
                raise ValueError(f"grid file {path}: cannot parse box {lines[q - 1]!r}")
            lo = [int(v) for v in m.group(1).split(",")]
            hi = [int(v) for v in m.group(2).split(",")]
            if l >= len(ref_ratio):          # deeper than amr.max_level: read and dropped (Amr::readProbinFile: in_finest = min(in_finest, max_level))
                continue
            r = ref_ratio[l]
            if len(lo) == 2 and slab is not None:   # 2-D grid file of a 2-D run lifted onto a slab: (x, y) -> (x, slab, z), the whole thickness
                ns_c = slab
                for _ in range(l):
                    ns_c *= ref_ratio[_]
                lo, hi = [lo[0], 0, lo[1]], [hi[0], ns_c - 1, hi[1]]
            boxes.append((tuple(v * r for v in lo), tuple((v + 1) * r - 1 for v in hi)))
        if l < len(ref_ratio):
            out.append(boxes)
    return out


def parse_text(text, table=None):
    """ParmParse tokenisation: `name = v1 v2 ...` (values up to the next `name =` or end of line; quotes group)"""
    table = {} if table is None else table
    for raw in text.splitlines():
        line = raw.split("#", 1)[0].strip()
        if not line:
            continue
        # several definitions may share a line
        parts = re.split(r"(?:(?<=\s)|^)([A-Za-z_][\w\.\-]*)\s*=", line)
        # parts: ['', name1, vals1, name2, vals2, ...]
        if len(parts) < 3:
            raise ValueError(f"cannot parse inputs line: {raw!r}")
        for i in range(1, len(parts) - 1, 2):
            vals = re.findall(r'"[^"]*"|\S+', parts[i + 1])
            table[parts[i]] = [v.strip('"') for v in vals]
    return table


class Inputs:
    def __init__(self, files=(), overrides=()):
        self.table = {}
        self.files = list(files)
        for f in files:
            with open(f) as fh:
                parse_text(fh.read(), self.table)
        for o in overrides:
            parse_text(o, self.table)
        self.used = set()
        self.ignored = []

    # ParmParse-style accessors ------------------------------------------------------------------------------------
    def has(self, k):
        return k in self.table

    def _get(self, k, n=None):
        self.used.add(k)
        v = self.table[k]
        if n is not None and len(v) < n:
            raise ValueError(f"inputs: {k} needs {n} values, has {len(v)}")
        return v

    def real(self, k, default=None):
        if k not in self.table:
            if default is None:
                raise KeyError(f"inputs: required key {k} is missing")
            return default
        return float(self._get(k, 1)[0])

    def integer(self, k, default=None):
        if k not in self.table:
            if default is None:
                raise KeyError(f"inputs: required key {k} is missing")
            return default
        v = self._get(k, 1)[0]
        if v.lower() in ("true", "t"):           # ParmParse reads booleans into ints
            return 1
        if v.lower() in ("false", "f"):
            return 0
        return int(v)

    def name_list(self, k, default):
        """amr.plot_vars / amr.derive_plot_vars (Amr::initPltAndChk): ALL, NONE or a list of names; absent: `default`"""
        v = list(self._get(k)) if k in self.table else [default]
        if len(v) == 1 and v[0] in ("ALL", "NONE"):
            return v[0]
        return v

    def string(self, k, default=None):
        if k not in self.table:
            return default
        return self._get(k, 1)[0]

    def reals(self, k, n, default=None):
        if k not in self.table:
            if default is None:
                raise KeyError(f"inputs: required key {k} is missing")
            return list(default)
        return [float(x) for x in self._get(k, n)[:n]]

    def ints(self, k, n, default=None):
        if k not in self.table:
            if default is None:
                raise KeyError(f"inputs: required key {k} is missing")
            return list(default)
        return [int(x) for x in self._get(k, n)[:n]]

    def refinement_indicators(self, max_level):
        """amr.refinement_indicators and their sub-keys (NavierStokes::error_setup, Source/NS_error.cpp:10-108) -> the arguments of
        Amr.set_regrid"""
        names = self.table.get("amr.refinement_indicators", [])
        self.used.add("amr.refinement_indicators")
        comps = {"x_velocity": 0, "y_velocity": 1, "z_velocity": 2, "density": 3, "tracer": 4, "mag_vort": -1}
        nxt = 5
        for flag, nm_ in (("ns.do_trac2", "tracer2"), ("ns.do_temp", "temp")):      # NavierStokes.cpp:43-48
            if self.integer(flag, 0):
                comps[nm_] = nxt
                nxt += 1
        rules = []
        for nm in names:
            pre = f"amr.{nm}."
            r = {}
            for key, mode in (("value_greater", 0), ("value_less", 1), ("vorticity_greater", 2), ("adjacent_difference_greater", 3)):
                if self.has(pre + key):
                    r["mode"] = mode
                    r["value"] = [float(v) for v in self._get(pre + key)]
            if "mode" not in r:
                raise NotImplementedError(f"inputs: refinement indicator {nm}: none of value_greater / value_less / vorticity_greater / "
                                          "adjacent_difference_greater given")
            if r["mode"] == 2:
                r["comp"] = -1
            else:
                fld = self.string(pre + "field_name")
                if fld not in comps:
                    raise NotImplementedError(f"inputs: refinement indicator {nm}: field {fld} is not available (state components and mag_vort are)")
                r["comp"] = comps[fld]
            if self.has(pre + "max_level"):
                r["max_level"] = self.integer(pre + "max_level")
            if self.has(pre + "in_box_lo"):
                r["box_lo"] = self.reals(pre + "in_box_lo", 3)
                r["box_hi"] = self.reals(pre + "in_box_hi", 3)
            for key in ("start_time", "end_time"):
                if self.has(pre + key):
                    raise NotImplementedError(f"inputs: refinement indicator {nm}: {key} is not implemented")
            rules.append(r)
        return dict(max_level=max_level, regrid_int=self.ints("amr.regrid_int", 1, [1])[0], rules=rules,
                    blocking_factor=self.ints("amr.blocking_factor", 1, [8])[0], max_grid_size=self.ints("amr.max_grid_size", 1, [32])[0],
                    grid_eff=self.real("amr.grid_eff", 0.7), n_error_buf=self.ints("amr.n_error_buf", 1, [1])[0],
                    compute_new_dt_on_regrid=self.integer("amr.compute_new_dt_on_regrid", 0),
                    do_refine_outflow=self.integer("ns.do_refine_outflow", 0), do_derefine_outflow=self.integer("ns.do_derefine_outflow", 1),
                    nbuf_outflow=self.integer("ns.Nbuf_outflow", 1))

    # two-dimensional inputs ---------------------------------------------------------------------------------------------
    def lift_2d(self):
        """A 2-D inputs file (amr.n_cell with two entries; AMREX_SPACEDIM == 2 builds of the reference, Exec/run2d) runs as a slab of the
        three-dimensional library: 2-D x -> x, 2-D y -> z (gravity and the hydrostatic outflow pressure act along the last coordinate in
        both, NavierStokesBase.cpp:3560, Projection.cpp:2000), the third direction y is periodic, `slab` cells thick with cubic cells,
        carries no flow and no variation.  The table is rewritten in place into the equivalent 3-D inputs; returns the slab thickness
        (cells on level 0) or None for a 3-D file.  The work is `slab` (8 for the default blocking factor) times that of a true 2-D build
        (DESIGN.md section 7, row J2)."""
        nc = self.table.get("amr.n_cell")
        if nc is None or len(nc) != 2:
            return None
        T = self.table
        nx, ny = int(nc[0]), int(nc[1])
        bf = int(T.get("amr.blocking_factor", ["8"])[0])
        # one blocking factor thick (the regrid lattice is three-dimensional).  The multigrid solvers coarsen the slab with the plane until
        # it is two cells thick and keep it at two from there on (IAMRX_MG_SLAB, set by iamr_amd.run for these runs; mlmg.hip
        # mg_slab_level) -- until round 4 the slab had to be 8 ... 32 cells thick for the hierarchy to reach <= 16 x 16 cells
        slab = max(8, bf)
        lo = [float(v) for v in T["geometry.prob_lo"][:2]]
        hi = [float(v) for v in T["geometry.prob_hi"][:2]]
        dx = (hi[0] - lo[0]) / nx
        T["amr.n_cell"] = [str(nx), str(slab), str(ny)]
        T["geometry.prob_lo"] = [repr(lo[0]), "0.0", repr(lo[1])]
        T["geometry.prob_hi"] = [repr(hi[0]), repr(slab * dx), repr(hi[1])]
        per = T.get("geometry.is_periodic", ["0", "0"])
        T["geometry.is_periodic"] = [per[0], "1", per[1]]
        for k in ("ns.lo_bc", "ns.hi_bc"):
            if k in T:
                T[k] = [T[k][0], "0", T[k][1]]
        for side in ("lo", "hi"):                               # the y faces become the z faces; velocities (u, v) -> (u, 0, v)
            for k in [k for k in list(T) if k.startswith(f"y{side}.")]:
                T[f"z{side}." + k.split(".", 1)[1]] = T.pop(k)
            for d in "xz":
                k = f"{d}{side}.velocity"
                if k in T:
                    T[k] = [T[k][0], "0.0", T[k][1]]
        for k in ("prob.blob_center", "prob.velocity_ic"):
            if k in T:
                T[k] = [T[k][0], "0.0", T[k][1]]
        for k in [k for k in T if k.endswith(".in_box_lo") or k.endswith(".in_box_hi")]:
            T[k] = [T[k][0], "-1.e200" if k.endswith("lo") else "1.e200", T[k][1]]
        if "amr.max_grid_size" in T and len(T["amr.max_grid_size"]) > 1:
            T["amr.max_grid_size"] = T["amr.max_grid_size"][:1]
        return slab

    def physical_bcs(self, per):
        """ns.lo_bc / ns.hi_bc (integers, the older style) or {x,y,z}{lo,hi}.type (strings) -> PhysBCType per face
        (NavierStokes::Initialize_bcs, NavierStokes.cpp:66-250); both given for a face must agree"""
        names = {"no_slip_wall": 5, "nsw": 5, "slip_wall": 4, "sw": 4, "mass_inflow": 1, "mi": 1, "pressure_outflow": 2, "po": 2, "symmetry": 3, "sym": 3}
        lo = self.ints("ns.lo_bc", 3) if self.has("ns.lo_bc") else [None] * 3
        hi = self.ints("ns.hi_bc", 3) if self.has("ns.hi_bc") else [None] * 3
        for d, nm in enumerate("xyz"):
            for side, arr in (("lo", lo), ("hi", hi)):
                key = f"{nm}{side}.type"
                if self.has(key):
                    s = self.string(key).lower()
                    if s in ("pressure_inflow", "pi"):
                        raise NotImplementedError(f"inputs: {key} = {s}: not implemented upstream either (NavierStokes.cpp:171-180)")
                    if s not in names:
                        raise ValueError(f"inputs: {key} = {s}: no valid BC type")
                    if per[d]:
                        raise ValueError(f"inputs: wrong BC type for the periodic boundary {nm}{side}")
                    if arr[d] is not None and arr[d] != names[s]:
                        raise ValueError(f"inputs: multiple conflicting BCs specified for {nm}{side}")
                    arr[d] = names[s]
                    if self.has(f"{nm}{side}.pressure") and self.real(f"{nm}{side}.pressure") != 0.0:
                        raise NotImplementedError("inputs: pressure outflow != 0 is not implemented upstream either (NavierStokes.cpp:193-196)")
                if arr[d] is None:
                    if per[d]:
                        arr[d] = 0
                    else:
                        raise KeyError(f"inputs: no valid BC type specified for {nm}{side}")
        return lo, hi

    # mapping ----------------------------------------------------------------------------------------------------------
    def problem(self):
        """-> dict(n, prob_lo, prob_hi, periodic, max_grid_size, params (kwargs of ns_params), prob (dict), max_step, stop_time)"""
        slab = self.lift_2d()
        max_level = self.integer("amr.max_level", 0)
        fine_boxes = []
        regrid = None
        if max_level > 0:
            # fixed refined grids only (amr.regrid_file, as Exec/run2d/test_grids/inputs_*): the tagging / clustering blocks exist
            # (iamrx_error_tag, iamrx_cluster_tags) but no regrid driver runs them during a run yet
            # amr.regrid_file: fixed grids for the whole run; amr.initial_grid_file: only the initial hierarchy comes from the file,
            # the run then regrids from amr.refinement_indicators every amr.regrid_int steps (upstream Amr::initialInit / Amr::regrid)
            gf = None
            initial_only = False
            if self.has("amr.regrid_file"):
                gf = self.string("amr.regrid_file")
            elif self.has("amr.initial_grid_file"):
                gf = self.string("amr.initial_grid_file")
                initial_only = True
            rr = self.ints("amr.ref_ratio", max_level, [2] * max_level)
            if any(r != 2 for r in rr):
                raise NotImplementedError(f"inputs: amr.ref_ratio = {rr}: only ratio 2 is implemented")
            if gf is not None:
                if not os.path.isabs(gf) and self.files:
                    gf = os.path.join(os.path.dirname(os.path.abspath(self.files[0])), gf)
                fine_boxes = read_grid_file(gf, rr, slab)[:max_level]
                if any(len(lo) != 3 for lev in fine_boxes for lo, hi in lev):
                    raise ValueError(f"grid file {gf}: two-dimensional boxes in a three-dimensional run")
                if initial_only:
                    regrid = self.refinement_indicators(max_level)
                    if not regrid["rules"]:
                        raise NotImplementedError("inputs: amr.initial_grid_file without amr.refinement_indicators: the grids would have to be "
                                                  "regenerated from error tags this run does not define (use amr.regrid_file for fixed grids)")
            else:
                regrid = self.refinement_indicators(max_level)
                if not regrid["rules"]:
                    raise NotImplementedError("inputs: amr.max_level > 0 needs amr.refinement_indicators or fixed grids (amr.regrid_file)")
        if max_level == 0 and self.has("amr.refinement_indicators"):
            self.refinement_indicators(1)              # a single-level run reads and drops the indicator definitions, as upstream does
        if self.integer("geometry.coord_sys", 0) != 0:
            raise NotImplementedError("inputs: only Cartesian coordinates (geometry.coord_sys = 0)")
        n = self.ints("amr.n_cell", 3)
        mgs = self.ints("amr.max_grid_size", 1, [32])[0]
        prob_lo = self.reals("geometry.prob_lo", 3)
        prob_hi = self.reals("geometry.prob_hi", 3)
        per = self.ints("geometry.is_periodic", 3, [0, 0, 0])
        lo_bc, hi_bc = self.physical_bcs(per)
        for d in range(3):
            if per[d] and (lo_bc[d] != 0 or hi_bc[d] != 0):
                raise ValueError("inputs: periodic direction with a non-Interior ns.lo_bc/hi_bc (NavierStokesBase.cpp:563-590)")
            if not per[d] and (lo_bc[d] not in (1, 2, 3, 4, 5) or hi_bc[d] not in (1, 2, 3, 4, 5)):
                raise NotImplementedError(f"inputs: ns.lo_bc/hi_bc = {lo_bc[d]}/{hi_bc[d]} in direction {d}: implemented are Interior (0), "
                                          "Inflow (1), Outflow (2), Symmetry (3), SlipWall (4) and NoSlipWall (5)")
        scheme = self.string("ns.advection_scheme", "Godunov_PLM")
        if scheme not in ("Godunov_PLM", "Godunov_PPM", "BDS"):      # NavierStokesBase.cpp:548-553
            raise NotImplementedError(f"inputs: ns.advection_scheme = {scheme}; Godunov_PLM, Godunov_PPM and BDS are implemented")
        for k in ("ns.do_LES", "particles.do_nspc_particles", "eb2.geom_type"):
            if self.has(k) and self.string(k) not in ("0", "all_regular"):
                raise NotImplementedError(f"inputs: {k} = {self.string(k)} is not implemented")
        do_trac2, do_temp = self.integer("ns.do_trac2", 0), self.integer("ns.do_temp", 0)
        ntrac = 2 if do_trac2 else 1
        # NavierStokes.cpp:268-282: n_scal_diff_coefs + n_temp_cond_coef must equal NUM_SCALARS - 1
        sdc = self.reals("ns.scal_diff_coefs", ntrac, [0.0] * ntrac)
        snames = ["density", "tracer"] + (["tracer2"] if do_trac2 else []) + (["temp"] if do_temp else [])
        p = dict(cfl=self.real("ns.cfl"), visc_coef=self.real("ns.vel_visc_coef", 0.0), tracer_diff_coef=sdc[0],
                 do_trac2=do_trac2, do_cons_trac2=self.integer("ns.do_cons_trac2", 0), tracer2_diff_coef=sdc[1] if do_trac2 else 0.0,
                 do_temp=do_temp, temp_cond_coef=self.real("ns.temp_cond_coef", 0.0) if do_temp else 0.0,
                 init_iter=self.integer("ns.init_iter", 2), init_vel_iter=self.integer("ns.init_vel_iter", 1),
                 init_shrink=self.real("ns.init_shrink", 1.0), change_max=self.real("ns.change_max", 1.1),
                 fixed_dt=self.real("ns.fixed_dt", -1.0), init_dt=self.real("ns.init_dt", -1.0), gravity=self.real("ns.gravity", 0.0),
                 be_cn_theta=self.real("ns.be_cn_theta", 0.5), visc_tol=self.real("ns.visc_tol", 1.0e-10),
                 use_forces_in_trans=self.integer("godunov.use_forces_in_trans", 0), do_mom_diff=self.integer("ns.do_mom_diff", 0), do_cons_trac=self.integer("ns.do_cons_trac", 0), use_ppm={"Godunov_PLM": 0, "Godunov_PPM": 1, "BDS": 2}[scheme],
                 do_denminmax=self.integer("ns.do_denminmax", 0), do_scalminmax=self.integer("ns.do_scalminmax", 0),
                 mac_tol=self.real("mac_proj.mac_tol", 1.0e-12), mac_abs_tol=self.real("mac_proj.mac_abs_tol", 1.0e-16),
                 proj_tol=self.real("proj.proj_tol", 1.0e-12), proj_abs_tol=self.real("proj.proj_abs_tol", 1.0e-16),
                 phys_lo=lo_bc, phys_hi=hi_bc)
        wlo, whi = [0.0] * 9, [0.0] * 9
        # NavierStokes.cpp:113-160: a no-slip wall takes the tangential components of its .velocity (the wall does not move along its
        # normal), a mass inflow all of them; slip walls, outflow and symmetry faces take none
        for d, name in enumerate("xyz"):
            for side, arr, bcs in (("lo", wlo, lo_bc), ("hi", whi, hi_bc)):
                if not self.has(f"{name}{side}.velocity"):
                    continue
                v = self.reals(f"{name}{side}.velocity", 3)
                if bcs[d] == 5:
                    v[d] = 0.0
                elif bcs[d] != 1:
                    v = [0.0, 0.0, 0.0]
                arr[3 * d:3 * d + 3] = v
        p["wall_vel_lo"], p["wall_vel_hi"] = wlo, whi
        # inflow values of the scalars: {x,y,z}{lo,hi}.density / .tracer / .tracer2 / .temp (NavierStokes::Initialize_bcs,
        # NavierStokes.cpp:66-170: defaults density 1, tracers 0, temp 1), [d*4 + slot]
        slo, shi = [0.0] * 12, [0.0] * 12
        for d, name in enumerate("xyz"):
            for q, sname in enumerate(snames):
                dflt = 1.0 if sname in ("density", "temp") else 0.0
                slo[4 * d + q] = self.real(f"{name}lo.{sname}", dflt) if self.has(f"{name}lo.{sname}") else dflt
                shi[4 * d + q] = self.real(f"{name}hi.{sname}", dflt) if self.has(f"{name}hi.{sname}") else dflt
            for sname in ("tracer2", "temp"):           # values of absent scalars are read and dropped upstream
                for side in ("lo", "hi"):
                    if sname not in snames and self.has(f"{name}{side}.{sname}"):
                        self.real(f"{name}{side}.{sname}", 0.0)
        p["scal_bc_lo"], p["scal_bc_hi"] = slo, shi
        probtype = self.integer("prob.probtype")
        if probtype == 1:
            prob = dict(probtype=1, rho0=1.0)
        elif probtype == 11:
            prob = dict(probtype=11, vfac=self.real("prob.velocity_factor", 0.0), a=self.real("prob.a", 1.0), b=self.real("prob.b", 1.0),
                        c=self.real("prob.c", 1.0), rho0=self.real("prob.density_ic", 1.0))
        elif probtype == 10:
            prob = dict(probtype=10, rho_1=self.real("prob.rho_1"), rho_2=self.real("prob.rho_2"), tra_1=self.real("prob.tra_1", 0.0),
                        tra_2=self.real("prob.tra_2", 0.0), pertamp=self.real("prob.perturbation_amplitude", 0.0),
                        interface_width=self.real("prob.interface_width", 1.0))
        elif probtype in (2, 4, 5, 6, 7):     # host-side initial data (iamr_amd/probinit.py)
            prob = dict(probtype=probtype, density_ic=self.real("prob.density_ic", 1.0), direction=self.integer("prob.direction", 0),
                        interface_width=self.real("prob.interface_width", 1.0), blob_radius=self.real("prob.blob_radius", 0.1),
                        blob_center=self.reals("prob.blob_center", 3, [0.0, 0.0, 0.0]), velocity_ic=self.reals("prob.velocity_ic", 3, [0.0, 0.0, 0.0]))
        else:
            raise NotImplementedError(f"inputs: prob.probtype = {probtype}; implemented: 1 (fluid at rest, LidDrivenCavity), 2 / 6 (bubble / hot spot), 4 (constant "
                                      "velocity + tracer blob), 5 (DoubleShearLayer), 7 (Euler), 10 (RayleighTaylor), 11 (TaylorGreen)")
        if slab:
            prob["dim"] = 2
        out = dict(n=n, prob_lo=prob_lo, prob_hi=prob_hi, periodic=per, max_grid_size=mgs, params=p, prob=prob, slab=slab,
                   max_step=self.integer("max_step", -1), stop_time=self.real("stop_time", -1.0),
                   plot_int=self.integer("amr.plot_int", -1), plot_file=self.string("amr.plot_file", "plt"), plot_vars=self.name_list("amr.plot_vars", "ALL"),
                   derive_plot_vars=self.name_list("amr.derive_plot_vars", "NONE"), fine_boxes=fine_boxes, regrid=regrid, max_level=max_level,
                   check_int=self.integer("amr.check_int", -1), check_file=self.string("amr.check_file", "chk"),
                   restart=self.string("amr.restart", "") if self.has("amr.restart") else "")
        for k, dflt in _UNIMPLEMENTED_UNLESS.items():
            if self.has(k) and self.string(k) != dflt:
                raise NotImplementedError(f"inputs: {k} = {self.string(k)} is not implemented (only {dflt})")
        for k in self.table:
            if k not in self.used:
                if k in _IGNORED_KEYS or k.startswith(_IGNORED_NAMESPACES):
                    self.ignored.append(k)
                else:
                    raise KeyError(f"inputs: key {k} is not understood by this library (not silently ignored)")
        return out
