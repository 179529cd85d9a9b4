"""`python -m iamr_amd.run <inputs file> [key=value ...]` -- IAMR's main loop (Source/main.cpp:60-145: ParmParse, Amr::init,
`while (step < max_step && time < stop_time) coarseTimeStep`) driving libiamrx.so from an unmodified IAMR inputs file (SURVEY row
f4): one level, or a hierarchy of fixed refined grids (amr.max_level > 0 with amr.regrid_file, subcycled, inviscid).  Under
torch.distributed.run a single-level run shards its boxes over the ranks (one process per GPU)."""
import os
import sys
import time


def init_level(ns, lay, lib, N, pr, n):
    """initial data of one level (prob_initData role, Source/prob/prob_init.cpp), n = the level's cell counts"""
    pb = pr["prob"]
    if pb["probtype"] == 1:
        ns.init_rest(pb["rho0"])
    elif pb["probtype"] in (2, 4, 5, 6, 7):
        from .probinit import set_initial_state
        ns.init_rest(pb["density_ic"])
        set_initial_state(ns, lay, lib, N, pb, n, pr["prob_lo"], pr["prob_hi"])
    elif pb["probtype"] == 10:
        ns.init_rayleightaylor(pb["rho_1"], pb["rho_2"], pb["tra_1"], pb["tra_2"], pb["pertamp"], pb["interface_width"])
    else:
        ns.init_taylorgreen(pb["vfac"], pb["a"], pb["b"], pb["c"], pb["rho0"])


def build_amr(pr, lib, N, world=1, **mg_kw):
    """hierarchy of fixed grids: level 0 chopped by amr.max_grid_size, the refined levels as the grid file gives them"""
    from .amr import Amr
    # 2-D run on its slab: the multigrid keeps the slab two cells thick (mlmg.hip mg_slab_level) -- a property of THIS problem's solvers,
    # handed over in their options (not process-wide tuning: ADVICE round 5)
    opts = lib.mg_opts(slab=1 if pr.get("slab") else 0, **mg_kw)
    g0 = lib.Geom.make(pr["n"], prob_lo=pr["prob_lo"], prob_hi=pr["prob_hi"], periodic=pr["periodic"])
    lays = [lib.Layout.decompose(tuple(pr["n"]), pr["max_grid_size"], world)] + [lib.Layout(b, [q % world for q in range(len(b))]) for b in pr["fine_boxes"]]
    amr = Amr(g0, lays, N.ns_params(**pr["params"]), opts)
    for l, lev in enumerate(amr.levels):
        init_level(lev, lays[l], lib, N, pr, [v * 2 ** l for v in pr["n"]])
    if pr.get("regrid"):
        # Amr::bldFineLevels: the initial hierarchy from the tags of the initial data, one level at a time; the new level's data are
        # the problem's initial data (initData), not the interpolant
        amr.set_regrid(**pr["regrid"])
        # amr.initial_grid_file: the initial hierarchy is the file's, the tags only drive the regrids during the run
        for _ in range(0 if pr["fine_boxes"] else pr["regrid"]["max_level"]):
            nl = amr.nlev
            if not amr.regrid():
                break
            for l in range(1, amr.nlev):
                init_level(amr.levels[l], amr.layouts[l], lib, N, pr, [v * 2 ** l for v in pr["n"]])
            if amr.nlev == nl and amr.nlev > pr["regrid"]["max_level"]:
                break
    return amr, amr.layouts, g0


def level_arrays(ns, lay, N, derived=()):
    """valid-region arrays of the level's boxes: the state components, then the derived quantities asked for (NavierStokes::derive)"""
    import numpy as np
    S = ns.data(N.NavierStokes.S_NEW)
    D = [ns.derive(name) for name in derived]
    boxes, arrs = [], []
    for li in range(S.nlocal()):
        a, lo = S.to_numpy(li)
        blo, bhi, gi = lay.local_box(li)
        ng = (a.shape[0] - (bhi[0] - blo[0] + 1)) // 2
        v = a[ng:a.shape[0] - ng, ng:a.shape[1] - ng, ng:a.shape[2] - ng, :]
        arrs.append(np.concatenate([v] + [d.to_numpy(li)[0] for d in D], axis=-1) if D else v.copy())
        boxes.append((tuple(blo), tuple(bhi)))
    return boxes, arrs


def plot_names_and_filter(pr):
    """(names of everything level_arrays returns for this run, derived names, predicate name -> written?): amr.plot_vars picks among the
    state variables, amr.derive_plot_vars adds derived ones (plotfile.plot_selection)"""
    from .plotfile import state_names, plot_selection
    state = state_names(pr["params"].get("do_trac2", 0), pr["params"].get("do_temp", 0))
    pv = pr.get("plot_vars", "ALL")
    if pr.get("slab") and isinstance(pv, list):          # a 2-D inputs file names the plane's velocities x_velocity, y_velocity: y is the slab's z
        pv = ["z_velocity" if v == "y_velocity" else v for v in pv]
    keep, der = plot_selection(state, pv, pr.get("derive_plot_vars", "NONE"))
    kept = {state[q] for q in keep} | set(der)
    return state + der, der, kept


def select_components(names, arrs, kept, renamed=None):
    """drop the components amr.plot_vars does not name; `renamed`: names after _plane's renaming, same order as `names`"""
    idx = [q for q, nm in enumerate(names) if nm in kept]
    out = renamed if renamed is not None else names
    return [out[q] for q in idx], [a[..., idx] for a in arrs]


def _plane(boxes, arrs, names):
    """the (x, z) plane of a 2-D run lifted onto a y-periodic slab (inputs.Inputs.lift_2d) as 2-D boxes / arrays / names: the boxes that
    start at y = 0, their first y-plane, the y-velocity (identically zero) dropped, z_velocity renamed"""
    import numpy as np
    keep = [q for q, (lo, hi) in enumerate(boxes) if lo[1] == 0]
    comps = [0, 2] + list(range(3, len(names)))
    # what makes the slab a 2-D run: no variation across it, no flow along it (checked on everything written; enforce_plane keeps it so)
    for a in arrs:
        scale = max(1.0, float(np.abs(a).max()))
        dev = max(float(np.abs(a - a[:, :1]).max()), float(np.abs(a[..., 1]).max()))
        if dev > 1e-8 * scale:
            raise RuntimeError(f"iamr_amd.run: the slab of a two-dimensional run lost its uniformity (deviation {dev:.3e}, scale {scale:.3e})")
    n2 = ["x_velocity", "y_velocity"] + list(names[3:])
    return ([((boxes[q][0][0], boxes[q][0][2]), (boxes[q][1][0], boxes[q][1][2])) for q in keep],
            [arrs[q][:, 0, :, :][..., comps].copy() for q in keep], n2)


def enforce_plane(levels, pr):
    """a two-dimensional run on its y-periodic slab (inputs.Inputs.lift_2d) after every coarse time step: the state, the pressure and its
    gradient become their means across the slab, the velocity / gradient component along it zero.  Upstream's 2-D build has no third
    direction to perturb; here the multigrid smoothers' (i + j + k) colouring leaves iteration error at the level of the solver
    tolerances that varies across the slab, which a flow that amplifies perturbations would grow into real three-dimensional motion over
    many steps.  Projecting it out each step keeps the run the 2-D run the inputs file describes (deviations removed are of the size of
    the solver tolerances; `_plane` still checks everything written).  Single-rank runs (every box is local)."""
    import numpy as np
    for l, lev in enumerate(levels):
        n = [v * 2 ** l for v in pr["n"]]
        for which, along in ((lev.S_NEW, 1), (lev.P_NEW, None), (lev.GP_NEW, 1)):
            mf = lev.data(which)
            G = mf.gather_valid(n)
            plane = G[:, :n[1]].mean(axis=1)                     # nodal in y: node n[1] is the periodic image of node 0
            if along is not None:
                plane[..., along] = 0.0
            for li in range(mf.nlocal()):
                a, lo = mf.to_numpy(li)
                blo, bhi, _ = mf.layout.local_box(li)
                vx = slice(blo[0] - lo[0], bhi[0] + mf.typ[0] - lo[0] + 1)
                vz = slice(blo[2] - lo[2], bhi[2] + mf.typ[2] - lo[2] + 1)
                vy = slice(blo[1] - lo[1], bhi[1] + mf.typ[1] - lo[1] + 1)
                a[vx, vy, vz] = plane[blo[0]:bhi[0] + mf.typ[0] + 1, None, blo[2]:bhi[2] + mf.typ[2] + 1]
                mf.from_numpy(a, li)
            lev.set_data(which, mf)


def write_plot_amr(amr, lays, pr, N, step, root):
    """NavierStokesBase::writePlotFile role for the hierarchy: one AMReX plotfile with every level (the five state components)"""
    from .plotfile import PlotFile, Level
    names, der, kept = plot_names_and_filter(pr)
    levels = []
    dts = amr.dts()
    for l, lev in enumerate(amr.levels):
        n = [v * 2 ** l for v in pr["n"]]
        dx = [(pr["prob_hi"][d] - pr["prob_lo"][d]) / n[d] for d in range(3)]
        boxes, arrs = level_arrays(lev, lays[l], N, der)
        if pr.get("slab"):
            boxes, arrs, names2 = _plane(boxes, arrs, names)
            plane_names = [nm for q, nm in enumerate(names) if q != 1]
            out_names, arrs = select_components(plane_names, arrs, kept, names2)
            levels.append(Level(((0, 0), (n[0] - 1, n[2] - 1)), [dx[0], dx[2]], boxes, arrs, step * 2 ** l, amr.time))
        else:
            out_names, arrs = select_components(names, arrs, kept)
            levels.append(Level(((0, 0, 0), tuple(v - 1 for v in n)), dx, boxes, arrs, step * 2 ** l, amr.time))
    path = f"{root}{step:05d}"
    if pr.get("slab"):
        PlotFile(out_names, amr.time, [pr["prob_lo"][0], pr["prob_lo"][2]], [pr["prob_hi"][0], pr["prob_hi"][2]], levels).write(path)
    else:
        PlotFile(out_names, amr.time, pr["prob_lo"], pr["prob_hi"], levels).write(path)
    return path


def main_amr(pr, inp, lib, N, rank=0, world=1):
    """hierarchy run; world > 1: the boxes of every level are spread over the ranks (level 0 by Layout.decompose, fixed refined grids
    round-robin, regridded levels by the library's knapsack), plotfiles are written by single-rank runs only"""
    say = print if rank == 0 else (lambda *a, **k: None)
    check_int, check_root = (pr.get("check_int", -1), pr.get("check_file", "chk")) if world == 1 else (-1, "chk")
    plot_int, plot_root = pr.get("plot_int", -1), pr.get("plot_file", "plt")
    if world > 1:
        plot_int = -1
    if pr.get("restart"):
        # amr.restart (Amr::restart): grids, data, times and step counters come from the checkpoint, parameters from the inputs file
        from . import checkpoint
        if world > 1:
            raise NotImplementedError("iamr_amd.run: amr.restart runs on one rank")
        g0 = lib.Geom.make(pr["n"], prob_lo=pr["prob_lo"], prob_hi=pr["prob_hi"], periodic=pr["periodic"])
        amr = checkpoint.restart(pr["restart"], g0, N.ns_params(**pr["params"]), stop_time=pr["stop_time"])
        if pr.get("regrid"):
            amr.set_regrid(**pr["regrid"])
        lays = amr.layouts
        step = checkpoint.read_header(pr["restart"])["level_steps"][0]
        say(f"RESTART from {pr['restart']}: step {step}, time {amr.time:.12g}, levels {amr.nlev}")
    else:
        amr, lays, g0 = build_amr(pr, lib, N, world)
        amr.post_init(pr["stop_time"])
        step = 0
        if plot_int > 0:
            say("PLOTFILE:", write_plot_amr(amr, lays, pr, N, 0, plot_root))
    if inp.ignored:
        say("inputs: ignored (I/O / verbosity / AMR bookkeeping) keys:", " ".join(sorted(inp.ignored)))
    t0 = time.perf_counter()
    while (pr["max_step"] < 0 or step < pr["max_step"]) and (pr["stop_time"] < 0 or amr.time < pr["stop_time"] - 1e-14):
        if pr["max_step"] < 0 and pr["stop_time"] < 0:
            break
        dt = amr.coarse_step()
        if pr.get("slab"):
            enforce_plane(amr.levels, pr)
        lays = amr.layouts                      # a regrid during the step replaces them
        step += 1
        say(f"STEP = {step} TIME = {amr.time:.12g} DT = {dt:.12g} LEVELS = {amr.nlev} GRIDS = {[len(l.boxes) for l in lays]}")
        if plot_int > 0 and step % plot_int == 0:
            say("PLOTFILE:", write_plot_amr(amr, lays, pr, N, step, plot_root))
        if check_int > 0 and step % check_int == 0:
            from . import checkpoint
            say("CHECKPOINT:", checkpoint.write(amr, check_root, step, pr.get("max_level")))
    lib.sync()
    say(f"Run time = {time.perf_counter() - t0:.6f}")
    return 0


def build(inp, lib, N, nranks=1, pr=None):
    pr = pr if pr is not None else inp.problem()
    opts = lib.mg_opts(slab=1 if pr.get("slab") else 0)        # (see build_amr)
    g = lib.Geom.make(pr["n"], prob_lo=pr["prob_lo"], prob_hi=pr["prob_hi"], periodic=pr["periodic"])
    lay = lib.Layout.decompose(tuple(pr["n"]), pr["max_grid_size"], nranks)
    ns = N.NavierStokes(g, lay, N.ns_params(**pr["params"]), opts)
    pb = pr["prob"]
    if pb["probtype"] == 1:
        ns.init_rest(pb["rho0"])
    elif pb["probtype"] in (2, 4, 5, 6, 7):
        from .probinit import set_initial_state
        ns.init_rest(pb["density_ic"])
        set_initial_state(ns, lay, lib, N, pb, pr["n"], pr["prob_lo"], pr["prob_hi"])
    elif pb["probtype"] == 10:
        ns.init_rayleightaylor(pb["rho_1"], pb["rho_2"], pb["tra_1"], pb["tra_2"], pb["pertamp"], pb["interface_width"])
    else:
        ns.init_taylorgreen(pb["vfac"], pb["a"], pb["b"], pb["c"], pb["rho0"])
    return ns, lay, g, pr


def write_plot(ns, lay, pr, N, step, root):
    """NavierStokesBase::writePlotFile role (single level): AMReX-format plotfile <root><step:05d> with the state variables of
    amr.plot_vars and the derived quantities of amr.derive_plot_vars.  Single-rank runs only (every box is local)."""
    from .plotfile import from_level_data
    names, der, kept = plot_names_and_filter(pr)
    boxes, arrs = level_arrays(ns, lay, N, der)
    path = f"{root}{step:05d}"
    if pr.get("slab"):
        boxes, arrs, names2 = _plane(boxes, arrs, names)
        out_names, arrs = select_components([nm for q, nm in enumerate(names) if q != 1], arrs, kept, names2)
        from_level_data((pr["n"][0], pr["n"][2]), (pr["prob_lo"][0], pr["prob_lo"][2]), (pr["prob_hi"][0], pr["prob_hi"][2]), boxes, arrs, ns.time, step, names=out_names).write(path)
    else:
        out_names, arrs = select_components(names, arrs, kept)
        from_level_data(tuple(pr["n"]), tuple(pr["prob_lo"]), tuple(pr["prob_hi"]), boxes, arrs, ns.time, step, names=out_names).write(path)
    return path


def main(argv):
    from .inputs import Inputs
    files = [a for a in argv if "=" not in a]
    over = [a for a in argv if "=" in a]
    if not files:
        print(__doc__)
        return 2
    inp = Inputs(files, over)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    from . import lib
    from . import ns as N
    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    lib.init(local_rank)
    if world > 1:
        from . import comm
        comm.init_rccl_from_torch(dist)
    pr = inp.problem()
    if pr.get("slab") and world > 1:
        raise NotImplementedError("iamr_amd.run: two-dimensional inputs (run on a y-periodic slab) are single-rank runs")
    if pr["fine_boxes"] or pr.get("regrid") or (pr.get("restart") and pr.get("max_level", 0) > 0):
        return main_amr(pr, inp, lib, N, rank, world)
    plot_int, plot_root = pr.get("plot_int", -1), pr.get("plot_file", "plt")
    check_int, check_root = (pr.get("check_int", -1), pr.get("check_file", "chk")) if world == 1 else (-1, "chk")
    if pr.get("restart"):
        from . import checkpoint
        if world > 1:
            raise NotImplementedError("iamr_amd.run: amr.restart runs on one rank")
        g = lib.Geom.make(pr["n"], prob_lo=pr["prob_lo"], prob_hi=pr["prob_hi"], periodic=pr["periodic"])
        ns = checkpoint.restart(pr["restart"], g, N.ns_params(**pr["params"]), single_level=True, stop_time=pr["stop_time"])
        lay = ns.layout
        step = checkpoint.read_header(pr["restart"])["level_steps"][0]
        print(f"RESTART from {pr['restart']}: step {step}, time {ns.time:.12g}")
    else:
        ns, lay, g, pr = build(inp, lib, N, world, pr)
        ns.post_init(pr["stop_time"])
        step = 0
        if plot_int > 0 and world == 1:
            print("PLOTFILE:", write_plot(ns, lay, pr, N, 0, plot_root))
    if rank == 0 and inp.ignored:
        print("inputs: ignored (I/O / verbosity / AMR bookkeeping) keys:", " ".join(sorted(inp.ignored)))
    t0 = time.perf_counter()
    while (pr["max_step"] < 0 or step < pr["max_step"]) and (pr["stop_time"] < 0 or ns.time < pr["stop_time"] - 1e-14):
        if pr["max_step"] < 0 and pr["stop_time"] < 0:
            break
        dt = ns.step()
        if pr.get("slab"):
            enforce_plane([ns], pr)
        step += 1
        if rank == 0:
            print(f"STEP = {step} TIME = {ns.time:.12g} DT = {dt:.12g}")
        if plot_int > 0 and world == 1 and step % plot_int == 0:
            print("PLOTFILE:", write_plot(ns, lay, pr, N, step, plot_root))
        if check_int > 0 and step % check_int == 0:
            from . import checkpoint
            print("CHECKPOINT:", checkpoint.write(ns, check_root, step))
    lib.sync()
    if rank == 0:
        print(f"Run time = {time.perf_counter() - t0:.6f}")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
