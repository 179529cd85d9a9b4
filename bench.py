#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native IAMR hot path.

Metric (BASELINE.json): cells-advanced/sec (whole node) + MLMG V-cycle ms, 256^3 TaylorGreen.
A "step" = one full NavierStokes::advance (predict_velocity, MAC projection, velocity + scalar
advection, updates, Crank-Nicolson tensor diffusion solve, nodal level projection) of the
TaylorGreen 3-D problem (reference Tutorials/TaylorGreen/inputs.3d.taylorgreen: nu = 1e-4, cfl 0.7,
init_iter 2, Godunov_PLM), one 256^3 box per GPU (weak scaling: rank r owns box r of an N-box level).
Inputs are generated on the device (closed-form initial condition), so the timed region starts with
everything resident in HBM.

Usage:  python bench.py [--gpus N] [--steps K] [--warmup W] [--n 256] [--no-cpu-baseline]
For N > 1 launch with  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


PMC_FILE = "round6_pmc.json"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=int(os.environ.get("IAMRX_BENCH_N", "256")), help="cells per direction of the per-GPU box")
    ap.add_argument("--c", type=float, default=1.0, help="prob.c (1 = fully 3-D regtest default)")
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; value = median region (SURVEY 8d: median of 5)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-upstream-shape", action="store_true", help="skip the pass with the reference's multigrid cycle shape")
    ap.add_argument("--no-multibox", action="store_true", help="skip the single-GPU 8-box / 64-box runs of the same problem")
    ap.add_argument("--cpu-n", type=int, default=96)
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--amr-n", type=int, default=256, help="base-level cells per direction of the secondary 2-level AMR workload (0: skip)")
    ap.add_argument("--amr-steps", type=int, default=3)
    ap.add_argument("--c3-n", type=int, default=512, help="base cells per direction of the secondary DoubleShearLayer 2D workload (config C3; 0: skip)")
    ap.add_argument("--c3-steps", type=int, default=3, help="timed coarse steps of the C3 workload")
    ap.add_argument("--rt-n", type=int, default=256, help="base cells per direction of the secondary RayleighTaylor workload (config C5: 3 levels; 0: skip)")
    ap.add_argument("--rt-steps", type=int, default=2, help="timed coarse steps of the C5 workload (after 4 coarse steps that build the 3 levels)")
    ap.add_argument("--ldc-steps", type=int, default=4, help="timed steps of the secondary LidDrivenCavity workload at --n^3 (single GPU; 0: skip)")
    ap.add_argument("--no-shard-proxy", action="store_true", help="skip the single-GPU proxies of the per-GPU work of an 8-GPU run (8 boxes of --n^3 kept as boxes)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="OpenMP threads of the oracle's smoother loops (the rest of the port is scalar); 0 = the CPUs this process may really use")
    return ap.parse_args()


def hip_event_time(lib, fn, reps):
    """average device time (ms) of fn() over reps launches, HIP events recorded on the library's stream"""
    L = lib.lib()
    for _ in range(2):
        fn()
    lib.sync()
    lib.check(L.iamrx_timer_start())
    for _ in range(reps):
        fn()
    ms = C.c_double()
    lib.check(L.iamrx_timer_stop(C.byref(ms)))
    return ms.value / reps


def kernel_rooflines(lib, n):
    """live per-kernel timings (HIP events on the launch stream) of the roofline-graded kernels at the bench size"""
    from iamr_amd import ns as N
    import numpy as np
    g = lib.Geom.make((n, n, n))
    lay = lib.Layout.single((n, n, n))
    cells = float(n) ** 3
    out = {}
    # ABec GSRB red+black sweep, variable b: 80 B/cell algorithmic
    b = [lib.MultiFab(lay, lib.face(d), 1, 0) for d in range(3)]
    for m in b:
        m.setval(1.0)
    phi = lib.MultiFab(lay, lib.CELL, 1, 1)
    rhs = lib.MultiFab(lay, lib.CELL, 1, 0)
    phi.setval(0.5)
    rhs.setval(1.0)
    t = hip_event_time(lib, lambda: (lib.abec_gsrb(g, 0.0, 1.0, None, b, phi, rhs, 0), lib.abec_gsrb(g, 0.0, 1.0, None, b, phi, rhs, 1)), 20)
    out["abec_gsrb_sweep"] = {"ms": t, "alg_bytes_per_cell": 80, "GBps": 80 * cells / t / 1e6}
    res = lib.MultiFab(lay, lib.CELL, 1, 0)
    t = hip_event_time(lib, lambda: lib.abec_residual(g, 0.0, 1.0, None, b, res, phi, rhs), 20)
    out["abec_residual"] = {"ms": t, "alg_bytes_per_cell": 48, "GBps": 48 * cells / t / 1e6}
    del b, phi, rhs, res
    # nodal 27-pt Gauss-Seidel sweep (8 colours): 32 B/node algorithmic
    sig = lib.MultiFab(lay, lib.CELL, 1, 4)
    sig.setval(1.0)
    x = lib.MultiFab(lay, lib.NODE, 1, 4)
    r = lib.MultiFab(lay, lib.NODE, 1, 4)
    x.setval(0.25)
    r.setval(1.0)
    nodes = float(n + 1) ** 3
    # one full Gauss-Seidel sweep incl. its ghost fills: plane-fused (2 passes) and reference form (8 colour passes)
    t = hip_event_time(lib, lambda: N.nodal_gs_sweep(g, x, r, sig, 1), 10)
    out["nodal_gs_sweep"] = {"ms": t, "alg_bytes_per_node": 32, "GBps": 32 * nodes / t / 1e6}
    t = hip_event_time(lib, lambda: N.nodal_gs_sweep(g, x, r, sig, 3), 10)     # the two kernel launches alone
    out["nodal_gs4_launch"] = {"ms": t / 2, "alg_bytes_per_launch": 16 * nodes, "GBps": 16 * nodes / (t / 2) / 1e6}
    t = hip_event_time(lib, lambda: N.nodal_gs_sweep(g, x, r, sig, 0), 5)
    out["nodal_gs_sweep_8pass"] = {"ms": t, "alg_bytes_per_node": 32, "GBps": 32 * nodes / t / 1e6}
    del sig, x, r
    # Godunov: ExtrapVelToFaces 72 B/cell, ComputeAofs velocity 104 B/cell
    vel = lib.MultiFab(lay, lib.CELL, 3, 3)
    frc = lib.MultiFab(lay, lib.CELL, 3, 1)
    um = [lib.MultiFab(lay, lib.face(d), 1, 1) for d in range(3)]
    vel.setval(0.3)
    frc.setval(0.1)
    dt = 0.3 / n
    t = hip_event_time(lib, lambda: lib.godunov_extrap_vel_to_faces(g, vel, frc, um, dt), 10)
    out["extrap_vel_to_faces"] = {"ms": t, "alg_bytes_per_cell": 72, "GBps": 72 * cells / t / 1e6}
    aofs = lib.MultiFab(lay, lib.CELL, 5, 0)
    divu = lib.MultiFab(lay, lib.CELL, 1, 1)
    divu.setval(0.0)
    t = hip_event_time(lib, lambda: lib.godunov_compute_aofs(g, aofs, 0, vel, 3, frc, divu, um, (0, 0, 0), dt, None, 1, 0), 10)
    out["compute_aofs_vel"] = {"ms": t, "alg_bytes_per_cell": 104, "GBps": 104 * cells / t / 1e6}
    S5 = lib.MultiFab(lay, lib.CELL, 5, 3)
    f5 = lib.MultiFab(lay, lib.CELL, 5, 1)
    S5.setval(0.3)
    f5.setval(0.1)
    t = hip_event_time(lib, lambda: lib.godunov_compute_aofs(g, aofs, 0, S5, 5, f5, divu, um, (0, 0, 0, 1, 0), dt, None, 1, 0), 10)
    out["compute_aofs_state5"] = {"ms": t, "alg_bytes_per_cell": 152, "GBps": 152 * cells / t / 1e6}
    return out


def usable_cpus():
    """CPUs this process can actually run on: affinity mask capped by the cgroup CPU quota (a 256-core box with a 16-CPU quota
    must not get 256 spinning OpenMP threads)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 32))


def cpu_baseline(n, steps, threads=1):
    """the CPU oracle (C port of the reference algorithm; its smoother loops are OpenMP-parallel, the rest is scalar) timed on a
    bounded sample with all host cores the OpenMP runtime offers"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    L = orc.lib()
    L.orc_set_threads(int(threads))
    g = orc.geom((n, n, n))
    p = orc.CNsParams()
    L.orc_ns_default_params(C.byref(p))
    p.cfl = 0.7
    p.visc_coef = 1e-4
    p.init_iter = 2
    o = orc.mg_opts()
    s = C.c_void_p(L.orc_ns_create(C.byref(g), C.byref(p), C.byref(o)))
    L.orc_ns_init_taylorgreen(s, C.c_double(1.0), C.c_double(1.0), C.c_double(1.0), C.c_double(1.0), C.c_double(1.0))
    L.orc_ns_post_init(s, C.c_double(-1.0))
    t0 = time.perf_counter()
    for _ in range(steps):
        L.orc_ns_step(s)
    dt = time.perf_counter() - t0
    L.orc_ns_destroy(s)
    return {"value": n ** 3 * steps / dt, "unit": "cells-advanced/s", "cores": threads, "kind": "port",
            "sample": f"TaylorGreen {n}^3 (same physics/settings), {steps} timed step(s) after post_init, oracle/liborc.so C port "
                      f"(OpenMP over planes in the Godunov, tensor and multigrid smoother / operator loops, {threads} thread(s)); "
                      f"NOT the IAMR CPU build (its sources need AMReX / AMReX-Hydro, absent here)"}


def proc_grid(world):
    """1 x py x pz with py <= pz and py * pz = world, as square as possible (2 -> 1x1x2, 4 -> 1x2x2, 8 -> 1x2x4): the boxes are split in y
    and z only, so that a row of cells (x, the contiguous direction) is never cut -- the one-launch red + black sweep then wraps a periodic
    x inside its rows as on a single box (k_abec_gsrb_rb<.., NBR, XO = false>), no ghost columns with one useful double per cache line are
    exchanged, and a box has 4 faces towards other GPUs instead of 6 (every pair of GPUs of the node has its own xGMI link)"""
    best = (1, 1, world)
    for py in range(1, world + 1):
        if world % py:
            continue
        pz = world // py
        if py <= pz and pz - py < best[2] - best[1]:
            best = (1, py, pz)
    return best


def file_blob_sha(path):
    """git blob hash of a file (what `git hash-object` prints)"""
    import hashlib
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def shard_proxy_workload(lib, n, steps=2, ldc_steps=2):
    """What ONE GPU of an 8-GPU weak-scaling run computes, measured on one GPU: the main workload's layout of 8 ranks -- eight n^3 boxes in
    the bench's own 1 x 2 x 4 arrangement (proc_grid(8)), periodic TaylorGreen -- and config C4's (LidDrivenCavity (2n)^3 as 2 x 2 x 2 boxes
    of n^3: every box has three wall faces and three faces towards other boxes), both with the boxes KEPT as boxes (IAMRX_COALESCE = 0): every
    box takes the kernels a rank that owns it would take (ghost-filled nodal passes, the multi-box red + black sweep with its two-layer
    exchange, coarse multigrid levels agglomerated at the same size as in a multi-rank run), only the exchanges are local copies instead
    of xGMI messages.  ms_per_box_step = ms_per_step / 8 is the per-GPU compute time of the sharded step; its ratio to the one-box step
    (periodic: the headline; walls: lid_driven_cavity) is the weak-scaling efficiency the kernels allow before any message is sent."""
    from iamr_amd import ns as N
    from iamr_amd.inputs import Inputs
    from iamr_amd import run as R
    from iamr_amd import ns as NS
    out = {}
    coalesce_before = lib.tuning_get("COALESCE", 1)
    lib.tuning_set("COALESCE", 0)
    try:
        pg = proc_grid(8)
        ntot = tuple(n * pg[d] for d in range(3))
        g = lib.Geom.make(ntot, prob_hi=tuple(float(pg[d]) for d in range(3)))
        lay = lib.Layout.decompose(ntot, n)
        ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.7, visc_coef=1e-4, init_iter=2, init_shrink=1.0), lib.mg_opts())
        ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
        ns.post_init(-1.0)
        ns.step()
        lib.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            ns.step()
        lib.sync()
        ms = (time.perf_counter() - t0) / steps * 1e3
        sm, sn, sv = ns.stats()
        out["taylorgreen_8_boxes_1x2x4"] = {"workload": f"TaylorGreen, {ntot[0]}x{ntot[1]}x{ntot[2]} cells in 8 boxes of {n}^3 kept as boxes (the layout of --gpus 8)",
                                            "ms_per_step": ms, "ms_per_box_step": ms / 8, "cells_per_sec": float(n) ** 3 * 8 / ms * 1e3, "steps": steps,
                                            "mlmg_iters": [sm.iters, sn.iters, sv.iters], "mlmg_vcycle_ms": [sm.vcycle_ms, sn.vcycle_ms, sv.vcycle_ms]}
        del ns
        if ldc_steps > 0:
            n2 = 2 * n
            inp = Inputs([os.path.join(ROOT, "tests", "golden", "regtest.3d.lid_driven_cavity")],
                         [f"amr.n_cell={n2} {n2} {n2}", f"amr.max_grid_size={n}", f"max_step={ldc_steps + 1}", f"ns.init_dt={0.0140625 * 64 / n2}"])
            pr = inp.problem()
            ns, lay, g, pr = R.build(inp, lib, NS, 1, pr)
            ns.post_init(pr["stop_time"])
            ns.step()
            lib.sync()
            t0 = time.perf_counter()
            for _ in range(ldc_steps):
                ns.step()
            lib.sync()
            ms = (time.perf_counter() - t0) / ldc_steps * 1e3
            sm, sn, sv = ns.stats()
            out["lid_driven_cavity_8_boxes_2x2x2"] = {"workload": f"LidDrivenCavity (regtest.3d.lid_driven_cavity), {n2}^3 cells in 8 boxes of {n}^3 kept as boxes (config C4 at 8 GPUs)",
                                                      "ms_per_step": ms, "ms_per_box_step": ms / 8, "cells_per_sec": float(n2) ** 3 / ms * 1e3, "steps": ldc_steps,
                                                      "mlmg_iters": [sm.iters, sn.iters, sv.iters], "mlmg_vcycle_ms": [sm.vcycle_ms, sn.vcycle_ms, sv.vcycle_ms]}
            del ns
    finally:
        lib.tuning_set("COALESCE", coalesce_before)
    return out


def ldc_workload(lib, n, steps):
    """BASELINE config C4 per GPU: the reference's regtest.3d.lid_driven_cavity (no-slip / slip walls on every side, variable-density
    capable MAC projection with Neumann walls, tensor solve with Dirichlet walls) at n^3 on one GPU, init_dt scaled with the mesh; ms per
    step after two warm-up steps.  A secondary figure: the level kernels with domain walls instead of index wrap."""
    from iamr_amd import ns as NS, run as R
    from iamr_amd.inputs import Inputs
    inp = Inputs([os.path.join(ROOT, "tests", "golden", "regtest.3d.lid_driven_cavity")],
                 [f"amr.n_cell={n} {n} {n}", f"amr.max_grid_size={n}", f"max_step={steps + 2}", f"ns.init_dt={0.0140625 * 64 / n}"])
    pr = inp.problem()
    ns, lay, g, pr = R.build(inp, lib, NS, 1, pr)
    ns.post_init(pr["stop_time"])
    for _ in range(2):
        ns.step()
    lib.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        ns.step()
    lib.sync()
    ms = (time.perf_counter() - t0) / steps * 1e3
    sm, sn, sv = ns.stats()
    return {"workload": "LidDrivenCavity 3D (regtest.3d.lid_driven_cavity), one %d^3 box, walls on every side" % n, "ms_per_step": ms,
            "cells_per_sec": float(n) ** 3 / ms * 1e3, "steps": steps,
            "mlmg_iters": {"mac_cc": sm.iters, "nodal": sn.iters, "tensor_visc": sv.iters},
            "mlmg_vcycle_ms": {"mac_cc": sm.vcycle_ms, "nodal": sn.vcycle_ms, "tensor_visc": sv.vcycle_ms}}


def c3_workload(lib, n, steps):
    """BASELINE config C3: DoubleShearLayer 2D, n x n base + one refined level (ratio 2) that follows the vorticity of the shear layers,
    regridded at the start of every coarse step (regrid + SyncRegister path).  The library is three-dimensional: the 2-D inputs file runs as
    a y-periodic slab one blocking factor (8 cells) thick (iamr_amd/inputs.py lift_2d), whose multigrid levels keep the slab at two cells
    (IAMRX_MG_SLAB) -- the figure quotes 2-D cells: cell updates of the plane per second, the slab's 8x redundant work included in the time."""
    from iamr_amd import ns as NS, run as R
    from iamr_amd.inputs import Inputs
    inp = Inputs([os.path.join(ROOT, "tests", "golden", "inputs.2d.doubleshearlayer_c3")],
                 [f"amr.n_cell={n} {n}", f"max_step={steps + 1}", "proj.proj_tol=1.0e-10"])
    pr = inp.problem()
    amr, lays, g0 = R.build_amr(pr, lib, NS, 1)
    amr.post_init(pr["stop_time"])
    amr.coarse_step()
    lib.sync()
    cells2d = 0
    t0 = time.perf_counter()
    for _ in range(steps):
        amr.coarse_step()
        slab0 = pr["n"][1]
        cells2d += sum(b_npts(lo, hi) for lo, hi in amr.layouts[0].boxes) // slab0
        if amr.nlev > 1:
            cells2d += 2 * sum(b_npts(lo, hi) for lo, hi in amr.layouts[1].boxes) // (2 * slab0)
    lib.sync()
    el = time.perf_counter() - t0
    fine = sum(b_npts(lo, hi) for lo, hi in amr.layouts[1].boxes) // (2 * pr["n"][1]) if amr.nlev > 1 else 0
    return {"workload": "DoubleShearLayer 2D (config C3), %d^2 base + one refined level (ratio 2, vorticity tags, regrid every step) as a y-periodic slab of %d cells with slab multigrid levels"
                        % (n, pr["n"][1]), "ms_per_coarse_step": el / steps * 1e3, "cells2d_per_sec": cells2d / el, "steps": steps,
            "levels": amr.nlev, "grids": [len(l.boxes) for l in amr.layouts], "fine_level_cells2d": fine, "fine_level_cover": fine / float(4 * n * n)}


def rt_workload(lib, n, steps, warm=4, **mg_kw):
    """BASELINE config C5 per GPU: the reference's regtest.3d.rayleightaylor, unmodified but for amr.n_cell = n^3 (variable density, gravity 1e9,
    Godunov_PPM, do_mom_diff, do_cons_trac, slip walls in z, max_level 2 on the vorticity, regrid every 2nd coarse step, subcycled).  The
    hierarchy grows from the base level to three levels within `warm` coarse steps; then `steps` coarse steps are timed (regrids included, as
    the reference's Run time includes them).  cells advanced per coarse step = sum over levels of cells x 2^level sub-steps."""
    from iamr_amd import ns as NS, run as R
    from iamr_amd.inputs import Inputs
    inp = Inputs([os.path.join(ROOT, "tests", "golden", "regtest.3d.rayleightaylor")],
                 [f"amr.n_cell={n} {n} {n}", f"amr.max_grid_size={max(32, n // 2)}", f"max_step={warm + steps + 2}"])
    pr = inp.problem()
    amr, lays, g0 = R.build_amr(pr, lib, NS, 1, **mg_kw)
    amr.post_init(pr["stop_time"])
    for _ in range(warm):
        amr.coarse_step()
    lib.sync()
    cells = 0
    per_step = []
    import ctypes as C
    nm0 = C.c_size_t(0)
    lib.check(lib.lib().iamrx_alloc_count(C.byref(nm0)))
    t0 = time.perf_counter()
    for _ in range(steps):
        t1 = time.perf_counter()
        amr.coarse_step()
        lib.sync()
        per_step.append((time.perf_counter() - t1) * 1e3)
        cells += sum((2 ** l) * sum(b_npts(lo, hi) for lo, hi in amr.layouts[l].boxes) for l in range(amr.nlev))
    el = time.perf_counter() - t0
    nm1 = C.c_size_t(0)
    lib.check(lib.lib().iamrx_alloc_count(C.byref(nm1)))
    st, stm = amr.sync_stats()
    levels = amr.nlev
    grids = [len(l.boxes) for l in amr.layouts]
    lev_cells = [sum(b_npts(lo, hi) for lo, hi in l.boxes) for l in amr.layouts]
    amr.profile(1)
    for _ in range(2):
        amr.coarse_step()
    sec, lsec = amr.profile(0)
    names = ["predict_velocity", "mac_project", "advection", "updates", "viscous", "nodal_project"]
    sections = {"reflux": sec[0] / 2, "avg_down": sec[1] / 2, "mac_sync_solve": sec[2] / 2, "mac_sync_rest": sec[3] / 2, "level_sync": sec[4] / 2, "regrid": sec[5] / 2}
    for l in range(min(amr.nlev, 4)):
        sections[f"advance_level{l}"] = sec[8 + l] / 2
        sections[f"advance_level{l}_sections"] = {k: v / 2 for k, v in zip(names, lsec[l][:6])}
    return {"workload": "RayleighTaylor 3D (config C5; regtest.3d.rayleightaylor with amr.n_cell = %d^3): variable density, gravity, Godunov_PPM, do_mom_diff, "
                        "do_cons_trac, slip walls in z, 3 levels from vorticity tags (ratio 2, subcycled, regrid every 2nd coarse step), one GPU" % n,
            "ms_per_coarse_step": el / steps * 1e3, "ms_of_each_coarse_step": per_step, "cells_advanced_per_sec": cells / el, "coarse_steps": steps,
            "warmup_coarse_steps": warm, "device_mallocs_in_timed_steps": nm1.value - nm0.value, "levels": levels, "grids": grids, "cells_per_level": lev_cells,
            "sync_project_iters": st.iters, "mac_sync_iters": stm.iters, "sections_ms_per_coarse_step": sections}


def b_npts(lo, hi):
    return (hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1)


def amr_workload(lib, n0, steps, rank=0, world=1, dist=None, layout_gpus=None, keep=None):
    """secondary workload (north_star: 2-level AMR TaylorGreen).  Per GPU: an n0^3 box of the base level and one ratio-2 refined box over
    its central (n0/2)^3 coarse cells (n0^3 fine cells), subcycled; nu = 1e-4 as in Tutorials/TaylorGreen/inputs.3d.taylorgreen.  N GPUs:
    the base boxes in the same process grid as the main workload (weak scaling), rank r owns base box r and refined box r.
    cells advanced per coarse step and GPU = n0^3 + 2 * n0^3.
    layout_gpus (tests): build the layout of that many GPUs but spread it over the `world` ranks present (box q -> rank q % world);
    keep: a list that receives the hierarchy object (tests compare its state)."""
    from iamr_amd import ns as N
    from iamr_amd.amr import Amr
    nbox = layout_gpus if layout_gpus else world
    pg = proc_grid(nbox)
    ntot = tuple(n0 * pg[d] for d in range(3))
    g0 = lib.Geom.make(ntot, prob_hi=tuple(float(pg[d]) for d in range(3)))
    cb, fb = [], []
    for r in range(nbox):
        o = (r % pg[0], (r // pg[0]) % pg[1], r // (pg[0] * pg[1]))
        cb.append((tuple(o[d] * n0 for d in range(3)), tuple((o[d] + 1) * n0 - 1 for d in range(3))))
        fb.append((tuple(2 * o[d] * n0 + n0 // 2 for d in range(3)), tuple(2 * o[d] * n0 + n0 // 2 + n0 - 1 for d in range(3))))
    own = [q % world for q in range(nbox)]
    lays = [lib.Layout(cb, own), lib.Layout(fb, own)]
    amr = Amr(g0, lays, N.ns_params(cfl=0.7, visc_coef=1e-4, init_iter=2, init_shrink=1.0), lib.mg_opts())
    for l in range(2):
        amr.levels[l].init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
    amr.post_init()
    amr.coarse_step()

    def barrier():
        lib.sync()
        if world > 1:
            dist.barrier()

    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        amr.coarse_step()
    barrier()
    el = time.perf_counter() - t0
    if world > 1:
        import torch
        tt = torch.tensor([el], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    st, stm = amr.sync_stats()
    cells = float(n0) ** 3 * 3.0 * nbox
    if keep is not None:
        keep.append(amr)
    # section breakdown: separate, synchronised pass of two coarse steps (does not perturb the timed region)
    amr.profile(1)
    for _ in range(2):
        amr.coarse_step()
    sec, lsec = amr.profile(0)
    names = ["predict_velocity", "mac_project", "advection", "updates", "viscous", "nodal_project"]
    sections = {"reflux": sec[0] / 2, "avg_down": sec[1] / 2, "mac_sync_solve": sec[2] / 2, "mac_sync_rest": sec[3] / 2, "level_sync": sec[4] / 2}
    for l in range(amr.nlev):
        sections[f"advance_level{l}"] = sec[8 + l] / 2
        sections[f"advance_level{l}_sections"] = {k: v / 2 for k, v in zip(names, lsec[l][:6])}
    return {"workload": f"TaylorGreen 3D, 2 levels, per GPU: {n0}^3 base box + one {n0}^3 refined box over its centre (ratio 2, subcycled), "
                        f"{pg[0]}x{pg[1]}x{pg[2]} GPUs, nu = 1e-4, periodic; advance + reflux + avgDown + mac_sync (incl. viscous sync) + MLsyncProject per coarse step",
            "n_gpus": world, "boxes_per_level": nbox,
            "cells_advanced_per_sec": cells * steps / el, "ms_per_coarse_step": el / steps * 1e3, "coarse_steps": steps,
            "sync_project_iters": st.iters, "mac_sync_iters": stm.iters, "sections_ms_per_coarse_step": sections}


def multibox_workload(lib, n, steps=3):
    """the main workload's n^3 problem on ONE GPU chopped into 8, 64 and 512 boxes (the last = the reference's default amr.max_grid_size = 32 at
    256^3).  The level object merges the boxes a rank owns (mf.h: coalesce_layout, the default): the chopped level runs the single-box kernels.
    `boxes_kept` is the same run with the merging switched off (IAMRX_COALESCE = 0): what the ghost-exchange path between boxes costs (copy
    plans between colour passes, no index wrap, per-box tiles) -- still the per-GPU cost of any layout whose boxes do not tile rectangles"""
    from iamr_amd import ns as N
    out = {}

    def one(mg):
        g = lib.Geom.make((n,) * 3)
        lay = lib.Layout.decompose((n,) * 3, mg)
        ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.7, visc_coef=1e-4, init_iter=2, init_shrink=1.0), lib.mg_opts())
        ns.init_taylorgreen(1.0, 1.0, 1.0, 1.0, 1.0)
        ns.post_init(-1.0)
        ns.step()
        lib.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            ns.step()
        lib.sync()
        ms = (time.perf_counter() - t0) / steps * 1e3
        sm, sn, sv = ns.stats()
        return {"ms_per_step": ms, "cells_per_sec": float(n) ** 3 / ms * 1e3, "mlmg_iters": [sm.iters, sn.iters, sv.iters],
                "mlmg_vcycle_ms": [sm.vcycle_ms, sn.vcycle_ms, sv.vcycle_ms]}
    for parts in (2, 4, 8):
        mg = n // parts
        r = one(mg)
        if parts < 8:
            before = lib.tuning_get("COALESCE", 1)
            lib.tuning_set("COALESCE", 0)
            try:
                r["boxes_kept"] = one(mg)
            finally:
                lib.tuning_set("COALESCE", before)
        out[f"{parts ** 3}x{mg}^3"] = r
    return out


def transport_selftest(lib, rank, world):
    """one halo exchange + one reduction over the freshly initialised transport (8^3 box per rank, stacked in z like the
    workload): the ghost planes must hold the neighbour ranks' values.  Raises if the transport does not deliver."""
    import numpy as np
    n = 8
    boxes = [((0, 0, r * n), (n - 1, n - 1, (r + 1) * n - 1)) for r in range(world)]
    lay = lib.Layout(boxes, list(range(world)))
    g = lib.Geom.make((n, n, n * world), prob_hi=(1.0, 1.0, float(world)))
    mf = lib.MultiFab(lay, lib.CELL, 1, 1)
    mf.setval(float(rank + 1))
    mf.fill_boundary(g)
    a, _ = mf.to_numpy(0)
    lo_nb, hi_nb = (rank - 1) % world + 1, (rank + 1) % world + 1
    if not (np.all(a[1:-1, 1:-1, 0, 0] == lo_nb) and np.all(a[1:-1, 1:-1, -1, 0] == hi_nb) and np.all(a[1:-1, 1:-1, 1:-1, 0] == rank + 1)):
        raise RuntimeError("halo exchange self-test delivered wrong ghost values")
    if mf.norm0() != float(world):
        raise RuntimeError("all-reduce self-test delivered a wrong maximum")


def upstream_shape_pass(lib, N, g, lay, c, steps=3):
    """the same level advanced with the multigrid cycle amrex::MLMG / MLNodeLaplacian use as the reference drives them:
    nodal smoother 4 sweeps, 2 + 2 smooth calls; zero initial guess for the MAC and nodal solves (no warm start).  Reported beside the
    product's default cycle so that the V-cycle metric BASELINE.json names is comparable."""
    import statistics as st
    lib.tuning_set("WARM_START", 0)
    try:
        ns = N.NavierStokes(g, lay, N.ns_params(cfl=0.7, visc_coef=1e-4, init_iter=2, init_shrink=1.0), lib.mg_opts(nodal_sweeps=4, nodal_nu1=2, nodal_nu2=2))
        ns.init_taylorgreen(1.0, 1.0, 1.0, c, 1.0)
        ns.post_init(-1.0)
        ns.step()
        lib.sync()
        ms, its = {"mac_cc": [], "nodal": [], "tensor_visc": []}, {"mac_cc": [], "nodal": [], "tensor_visc": []}
        t0 = time.perf_counter()
        for _ in range(steps):
            ns.step()
            for k, s in zip(("mac_cc", "nodal", "tensor_visc"), ns.stats()):
                ms[k].append(s.vcycle_ms)
                its[k].append(s.iters)
        lib.sync()
        el = time.perf_counter() - t0
        del ns
    finally:
        lib.tuning_set("WARM_START", 1)
    return ({k: st.median(v) for k, v in ms.items()}, {k: st.median(v) for k, v in its.items()}, el / steps * 1e3)


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    # IAMRX_BENCH_TRANSPORT=gloo: all ranks share GPU 0 and talk through the host-staged gloo transport -- lets the N > 1 code path
    # of this script run on a 1-GPU box (tests/test_gpu_dist.py); the numbers of such a run mean nothing
    shared_gpu = world > 1 and os.environ.get("IAMRX_BENCH_TRANSPORT") == "gloo"
    tdev = "cpu" if shared_gpu else "cuda"
    if shared_gpu:
        local_rank = 0
        dist.init_process_group("gloo")
    elif world > 1:
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from iamr_amd import lib
    from iamr_amd import ns as N
    lib.init(local_rank)
    if shared_gpu:
        from iamr_amd import comm
        transport = "gloo-host-staged (forced, shared GPU)"
        comm.init_gloo_callback(dist)
        transport_selftest(lib, rank, world)
    elif world > 1:
        from iamr_amd import comm
        transport = "rccl"
        # N real GPUs: the RCCL transport (grouped ncclSend/ncclRecv + ncclAllReduce over xGMI) must come up, a scaling number measured
        # over host-staged gloo would mean nothing -- fail loudly instead (IAMRX_BENCH_TRANSPORT=gloo is the explicit test-only override)
        comm.init_rccl_from_torch(dist)
        transport_selftest(lib, rank, world)

    n = a.n
    # one n^3 box per GPU (weak scaling), the boxes split in y and z only (proc_grid) -- SURVEY 8(e)
    pgrid = proc_grid(world)
    ntot = tuple(n * pgrid[d] for d in range(3))
    boxes = []
    for r in range(world):
        ix, iy, iz = r % pgrid[0], (r // pgrid[0]) % pgrid[1], r // (pgrid[0] * pgrid[1])
        boxes.append(((ix * n, iy * n, iz * n), ((ix + 1) * n - 1, (iy + 1) * n - 1, (iz + 1) * n - 1)))
    lay = lib.Layout(boxes, list(range(world)))
    g = lib.Geom.make(ntot, prob_hi=tuple(float(pgrid[d]) for d in range(3)))
    params = N.ns_params(cfl=0.7, visc_coef=1e-4, init_iter=2, init_shrink=1.0)
    ns = N.NavierStokes(g, lay, params, lib.mg_opts())
    ns.init_taylorgreen(1.0, 1.0, 1.0, a.c, 1.0)
    ns.post_init(-1.0)
    # HIP events around every 7th finest-level launch of the two smoother kernels that lead the step's kernel time (k_abec_gsrb,
    # k_nodal_gs4; profiles/round3_kernel_stats.csv), on their launch stream: opened during the warm-up (creates the event pools),
    # re-opened for the timed region and read after it -> roofline.avg_ms is measured inside the timed steps
    probe_on = world == 1 and os.environ.get("IAMRX_BENCH_PROBE", "1") != "0"
    # stride 7: a V-cycle issues 8 finest-level colour passes (the first of them the cheaper zero-initial-guess form), so a stride of 8 would
    # time the same position of every cycle
    PROBES = {"gs4": (0, (n + 1) ** 3, 7), "gsrb": (1, n ** 3, 7), "god_z": (2, n ** 3, 1), "pred_z": (3, n ** 3, 1)}

    def probes_start():
        for which, pts, stride in PROBES.values():
            lib.check(lib.lib().iamrx_kernel_probe_start(which, C.c_long(pts), stride))

    def probes_stop():
        res = {}
        for name, (which, pts, _stride) in PROBES.items():
            ms, nl = C.c_double(), C.c_long()
            lib.check(lib.lib().iamrx_kernel_probe_stop(which, C.byref(ms), C.byref(nl)))
            res[name] = (ms.value / nl.value, nl.value) if nl.value > 0 else None
        return res

    if probe_on:
        probes_start()
    for _ in range(a.warmup):
        ns.step()
    if probe_on:
        probes_stop()
        probes_start()

    def barrier():
        lib.sync()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    mac_ms, nod_ms, visc_ms, mac_it, nod_it, visc_it = [], [], [], [], [], []
    stalled = [0, 0, 0]           # solves of the timed steps that ended on the round-off-floor exit (converged == 2)
    def nmalloc():
        v = C.c_size_t()
        lib.check(lib.lib().iamrx_alloc_count(C.byref(v)))
        return v.value

    def nsync():
        v = C.c_size_t()
        lib.check(lib.lib().iamrx_sync_count(C.byref(v)))
        return v.value

    # SURVEY 8d's protocol: `repeats` timed regions of EXACTLY `steps` steps, each bracketed by barrier + synchronize on both sides and reduced
    # with MAX over the ranks; the reported region is the median one
    def nexch():
        v = (C.c_size_t * 4)()
        lib.check(lib.lib().iamrx_exchange_counts(v))
        return list(v)

    m0 = nmalloc()
    s0 = nsync()
    x0 = nexch()
    regions = []
    for _rep in range(max(1, a.repeats)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            ns.step()
            sm, sn, sv = ns.stats()
            mac_ms.append(sm.vcycle_ms); nod_ms.append(sn.vcycle_ms); visc_ms.append(sv.vcycle_ms)
            mac_it.append(sm.iters); nod_it.append(sn.iters); visc_it.append(sv.iters)
            for q, stq in enumerate((sm, sn, sv)):
                stalled[q] += 1 if stq.converged == 2 else 0
        barrier()
        el_r = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el_r], dtype=torch.float64, device=tdev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el_r = float(tt.item())
        regions.append(el_r)
    insitu = probes_stop() if probe_on else {}      # name -> (mean duration in ms, number of launches timed inside the timed regions)
    gs4_insitu = insitu.get("gs4")
    gsrb_insitu = insitu.get("gsrb")
    nreg = len(regions)
    mallocs_in_loop = nmalloc() - m0
    syncs_in_loop = (nsync() - s0) / nreg
    x1 = nexch()
    nst = max(1, nreg * a.steps)
    # halo exchanges of this rank per step: issued on the main stream in front of the kernel that needs them (exposed) / on the side stream
    # beside interior work (hidden as far as that work lasts), iamrx_exchange_counts; N = 1: none
    exch = {"exposed_per_step": (x1[0] - x0[0]) / nst, "exposed_MB_per_step": (x1[1] - x0[1]) * 8 / 1e6 / nst,
            "overlapped_per_step": (x1[2] - x0[2]) / nst, "overlapped_MB_per_step": (x1[3] - x0[3]) * 8 / 1e6 / nst}
    el = sorted(regions)[nreg // 2]
    cells_total = float(n) ** 3 * world
    value = cells_total * a.steps / el

    out = None
    # the 2-level AMR workload is collective too: every rank runs it (base and refined level sharded over the ranks)
    amr_res = amr_workload(lib, a.amr_n, a.amr_steps, rank, world, dist if world > 1 else None) if (a.amr_n > 0 and a.amr_steps > 0) else None
    # per-section breakdown (separate, synchronised pass so it does not perturb the timed region); a time step is collective:
    # EVERY rank takes these two steps
    ns.profile(2)
    for _ in range(2):
        ns.step()
    sec = ns.profile(0)
    if rank == 0:
        import statistics as st
        kr = kernel_rooflines(lib, n) if world == 1 else {}
        # dominant kernel of the step by summed duration (profiles/round3_kernel_stats.csv): k_abec_gsrb, one colour pass of the
        # cell-centred GSRB smoother (MAC projection, scalar diffusion); second: k_nodal_gs4 (reported beside it)
        roofline = None
        cells = float(n) ** 3

        def pmc_traffic(key_substr, field="hbm_bytes_per_launch"):
            """HBM bytes per launch of the 256^3-level launches inside the step, from the PMC passes over this script (FETCH_SIZE / WRITE_SIZE,
            separate rocprofv3 runs, tools/collect_pmc.sh, corrected as calibrated in profiles/round1_pmc.json).  The counters describe ONE
            build of the kernels: the file records the git blob hashes of the kernel sources it was collected with; any other source =>
            traffic is unknown (null), never a stale figure"""
            try:
                pmc = json.load(open(os.path.join(ROOT, "profiles", PMC_FILE)))
                if n != 256:
                    return None
                for src, blob in pmc.get("source_blobs", {}).items():
                    if blob != file_blob_sha(os.path.join(ROOT, "iamr_amd", "csrc", src)):
                        return None
                hits = [v for k, v in pmc["kernels"].items() if key_substr in k]
                return (hits[0].get(field) if hits else None)
            except Exception:
                return None

        gsrb_iso = kr.get("abec_gsrb_sweep")
        roofline_abec = None
        if gsrb_iso:
            # one red + black sweep of the MAC solve's smoother in ONE launch (round 4: k_abec_gsrb_rb; rounds 1-3 timed one colour pass).
            # achieved / frac: against the kernel's OWN compulsory traffic -- it recomputes the face coefficients from the cell-centred
            # density (AbecCoef::sig) and so moves less than SURVEY 8d's 80 B/cell of the operation (phi 1 + 1, rhs 1, three face-coefficient
            # arrays ...): phi read 8 + written 8, rhs 8, density 8 = 32 B/cell per sweep.  The 8d figure is kept beside it (survey_8d_*; a
            # fraction above 1 there only says that the kernel does not move the arrays 8d counts).  `traffic` (PMC) is to be read against own.
            alg = 80.0 * cells
            own = 32.0 * cells
            ms = gsrb_insitu[0] if gsrb_insitu else gsrb_iso["ms"]
            gbps = own / ms / 1e6
            rw = 16 // max(1, n // 128)
            nty = (n + rw - 3) // (rw - 2)
            nch = max(1, 256 // nty); tzp = max(8, (n + nch - 1) // nch); nch = (n + tzp - 1) // tzp
            rb_grid = 8 * ((nty * nch + 7) // 8) * 1024
            roofline_abec = {"kernel": "k_abec_gsrb_rb<1, 16, false, false, false, false, false, false> (a red AND a black pass of the cell-centred GSRB smoother of the MAC projection in one "
                                       "out-of-place launch, face coefficients recomputed from the cell-centred density once per face; "
                                       "profiles/round6_final_kernel_stats.csv)", "bound": "hbm",
                             "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0,
                             "traffic": pmc_traffic("k_abec_gsrb_rb<1, 16, false, false, false, false, false, false> grid=%d" % rb_grid),
                             "algorithmic_bytes_per_launch": own, "avg_ms": ms,
                             "bytes_counted": "the kernel's own compulsory traffic, 32 B/cell per sweep: phi read 8 + written 8, rhs 8, density 8",
                             "survey_8d_bytes_per_launch": alg, "survey_8d_GBps": alg / ms / 1e6, "survey_8d_frac": alg / ms / 1e6 / 8000.0,
                             "launches_timed": gsrb_insitu[1] if gsrb_insitu else None,
                             "timing": "HIP events around every 7th finest-level launch inside the timed steps" if gsrb_insitu else "isolated loop",
                             "isolated_loop_ms_array_coefficients_two_colour_passes": gsrb_iso["ms"]}
        # the kernel with the largest summed duration of the step (profiles/round6_final_kernel_stats.csv: k_nodal_gsr over its launch grids, ahead of
        # k_abec_gsrb_rb and k_god_z) is the nodal Gauss-Seidel pass: it is `roofline`; the MAC sweep and the Godunov kernels are reported beside it
        dom = kr.get("nodal_gs4_launch")
        roofline_gs4 = None
        if dom:
            # duration: mean over the finest-level launches inside the timed steps (HIP events on the launch stream); the isolated
            # back-to-back loop of kernel_rooflines() is kept beside it (warm L2, no interleaved fills: shorter)
            ms = gs4_insitu[0] if gs4_insitu else dom["ms"]
            gbps = dom["alg_bytes_per_launch"] / ms / 1e6
            roofline_gs4 = {"kernel": "k_nodal_gsr<4> (4 of the 8 Gauss-Seidel colours of the nodal smoother per launch, register-resident planes, variable "
                                      "sigma; 16 B/node per launch = SURVEY 8d's 32 B/node per sweep)", "bound": "hbm",
                            "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0,
                            "traffic": pmc_traffic("k_nodal_gsr<4, true, false, false> grid=131072"),
                            "algorithmic_bytes_per_launch": dom["alg_bytes_per_launch"], "avg_ms": ms,
                            # a launch updates the planes of one z-parity and reads x and sigma of both: x 8 + sigma 8 + rhs 4 + output 4 B/node
                            "two_launch_design_minimum_bytes_per_launch": 24.0 * (n + 1) ** 3,
                            "two_launch_design_frac": 24.0 * (n + 1) ** 3 / ms / 1e6 / 8000.0,
                            "launches_timed": gs4_insitu[1] if gs4_insitu else None,
                            "timing": "HIP events around every 7th finest-level launch inside the timed steps" if gs4_insitu else "isolated loop",
                            "isolated_loop_ms": dom["ms"]}
        # the Godunov kernels are bound by the vector pipe, not by HBM: both fractions are reported.  valu_frac =
        # wave-level vector instructions per launch (SQ_INSTS_VALU, third PMC pass of tools/collect_pmc.sh) / duration / the issue peak
        # of 256 CUs x 4 SIMDs x 1 wave64 instruction per 4 cycles at 2.4 GHz
        VALU_PEAK = 256 * 4 * 2.4e9 / 4

        def godunov_roofline(name, probe, grid_key, alg_bytes, what):
            if not probe:
                return None
            ms = probe[0]
            gbps = alg_bytes / ms / 1e6
            valu = pmc_traffic(grid_key, "SQ_INSTS_VALU")
            return {"kernel": what, "bound": "valu", "achieved": gbps, "peak": 8000.0, "unit": "GB/s", "frac": gbps / 8000.0,
                    "traffic": pmc_traffic(grid_key), "algorithmic_bytes_per_launch": alg_bytes, "avg_ms": ms, "launches_timed": probe[1],
                    "timing": "HIP events around every launch inside the timed steps",
                    "valu_insts_per_launch": valu, "valu_frac": (valu / (ms * 1e-3) / VALU_PEAK) if valu else None,
                    "valu_peak_wave_insts_per_s": VALU_PEAK}

        # algorithmic bytes per cell: advection of the 5 state components in one launch = 5 x (state 8 + forcing 8 + aofs 8) + mac 24 + divu 8;
        # prediction = velocity 24 + forcing 24 + the three face velocities 24
        zwg = ((n + 13) // 14) ** 2 * ((n + 63) // 64)         # workgroups of one component (14 x 14 cells, 64-plane chunks), 256 threads each
        zgrid = (zwg + 7) // 8 * 8 * 256
        roofline_god = godunov_roofline("god_z", insitu.get("god_z"), "k_god_z<14, 14, 256, 2, false, false> grid=%d" % (5 * zgrid), 152.0 * cells,
                                        "k_god_z<14,14> (fused z-marching Godunov advection of the 5 state components: ComputeFluxesOnBoxFromState + "
                                        "ComputeDivergence / ComputeConvectiveTerm in one launch; 14 x 14 cells = a grown tile of 16 x 16 = 256 threads)")
        roofline_pred = godunov_roofline("pred_z", insitu.get("pred_z"), "k_pred_z<14, 14, 256, false, false> grid=%d" % zgrid, 72.0 * cells,
                                         "k_pred_z<14,14> (fused z-marching ExtrapVelToFaces)")
        ups = upstream_shape_pass(lib, N, g, lay, a.c) if (world == 1 and not a.no_upstream_shape) else None
        out = {
            "metric": "cells-advanced/sec", "value": value, "unit": "cells/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": el / a.steps * 1e3, "repeats": nreg, "ms_per_step_of_each_region": [r / a.steps * 1e3 for r in regions],
            "protocol": "median of `repeats` timed regions of `steps` steps each (SURVEY 8d), every region between barrier + synchronize, max over ranks",
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"TaylorGreen 3D single level, one {n}^3 box per GPU in a {pgrid[0]}x{pgrid[1]}x{pgrid[2]} arrangement ({ntot[0]}x{ntot[1]}x{ntot[2]} cells), periodic, "
                                   f"nu=1e-4 cfl=0.7 Godunov_PLM be_cn_theta=0.5, full NavierStokes::advance per step",
                       "cells": cells_total, "prob_c": a.c},
            "mlmg_vcycle_ms": {"mac_cc": st.median(mac_ms), "nodal": st.median(nod_ms), "tensor_visc": st.median(visc_ms)},
            "mlmg_iters": {"mac_cc": st.median(mac_it), "nodal": st.median(nod_it), "tensor_visc": st.median(visc_it)},
            # solves of the timed steps that stopped on the round-off floor of their residual (within 10x of the target, < 10 % gained over
            # three cycles: converged == 2, a warning on stderr) instead of reaching the reference's tolerance -- of nreg * steps solves each
            "mlmg_stalled_solves": {"mac_cc": stalled[0], "nodal": stalled[1], "tensor_visc": stalled[2], "of_solves_each": nreg * a.steps},
            "mlmg_vcycle_ms_upstream_shape": ups[0] if ups else None,
            "mlmg_iters_upstream_shape": ups[1] if ups else None,
            "ms_per_step_upstream_shape": ups[2] if ups else None,
            "sections_ms_per_step": {k: v / 2 for k, v in zip(["predict_velocity", "mac_project", "advection", "updates", "viscous", "nodal_project"], sec[:6])},
            "device_mallocs_in_timed_region": mallocs_in_loop,
            "host_syncs_per_step": syncs_in_loop / a.steps,
            "transport": transport if world > 1 else "none (single GPU)",
            "halo_exchanges_rank0": exch,
            "kernels": kr,
            "roofline": roofline_gs4,
            "roofline_abec_sweep": roofline_abec,
            "roofline_godunov_advection": roofline_god,
            "roofline_godunov_prediction": roofline_pred,
        }
        if amr_res is not None:
            out["amr"] = amr_res
        if world == 1 and not a.no_multibox:
            out["single_gpu_multibox"] = multibox_workload(lib, n)
            out["single_gpu_multibox"]["1x%d^3" % n] = {"ms_per_step": el / a.steps * 1e3, "cells_per_sec": value}
        traffic_source = ("PMC passes (FETCH_SIZE / WRITE_SIZE / SQ_INSTS_VALU, separate rocprofv3 runs of tools/collect_pmc.sh over this script on the builder's "
                          "lease) stored in profiles/%s and keyed by the git blob hashes of the kernel sources: not measured by this run; null when "
                          "a kernel source differs from the one the counters were collected with" % PMC_FILE)
        for key in ("roofline", "roofline_abec_sweep", "roofline_godunov_advection", "roofline_godunov_prediction"):
            if out.get(key):
                out[key]["traffic_source"] = traffic_source
        if world == 1 and not a.no_shard_proxy and n <= 256:
            try:
                out["shard_proxy"] = shard_proxy_workload(lib, n, ldc_steps=2 if a.ldc_steps > 0 else 0)
                out["shard_proxy"]["one_box_ms_per_step"] = el / a.steps * 1e3
            except Exception as e:
                out["shard_proxy"] = {"error": str(e)[:200]}
        if world == 1 and a.ldc_steps > 0:
            try:
                out["lid_driven_cavity"] = ldc_workload(lib, n, a.ldc_steps)
            except Exception as e:                      # a secondary figure must not take the bench line with it
                out["lid_driven_cavity"] = {"error": str(e)[:200]}
        if world == 1 and a.c3_n > 0:
            try:
                out["double_shear_layer_2d"] = c3_workload(lib, a.c3_n, a.c3_steps)
            except Exception as e:
                out["double_shear_layer_2d"] = {"error": str(e)[:200]}
        if world == 1 and a.rt_n > 0 and a.rt_steps > 0:
            try:
                out["rayleigh_taylor"] = rt_workload(lib, a.rt_n, a.rt_steps)
            except Exception as e:
                out["rayleigh_taylor"] = {"error": str(e)[:200]}
        if not a.no_cpu_baseline and world == 1:       # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(a.cpu_n, a.cpu_steps, a.cpu_threads if a.cpu_threads > 0 else usable_cpus())
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
