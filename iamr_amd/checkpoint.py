"""Checkpoint / restart of a run: amrex::Amr::checkPoint + AmrLevel::checkPoint + StateData::checkPoint as IAMR drives them
(NavierStokesBase::checkPoint / restart, Source/NavierStokesBase.cpp:856-897, 2684-2727; amr.check_int / amr.check_file / amr.restart,
Exec/run3d/regtest.3d.euler-restart).

Layout of a checkpoint directory <root><step:05d>/ (upstream AMReX's; its sources are not in the reference tree, so the text layout is
restated from the published format -- the field ORDER is Amr::checkPoint's, the MultiFab files are VisMF files exactly as the plotfile
writer produces them, iamr_amd/plotfile.py, which is pinned byte for byte on the reference's committed plotfiles):

  Header                      CheckPointVersion_1.0, dim, cumulative time, max_level, finest_level, geometry of every level, ref_ratio,
                              dt_level, dt_min, n_cycle, level_steps, level_count; then per level (AmrLevel::checkPoint): level, geometry,
                              the BoxArray, the number of state types and per state type (StateData::checkPoint) domain, BoxArray, old / new
                              time interval, the number of MultiFabs written (2: new + old, "dump_old") and their relative paths
  Level_<l>/SD_<t>_New_MF_H, _D_00000 / SD_<t>_Old_MF_*     State_Type (t = 0: u v w rho tracer, 1 ghost cell), Press_Type (1, nodal),
                              Gradp_Type (2, 3 comps)
  iamrx_restart.json          what this library keeps beyond upstream's StateData and needs for a BIT-IDENTICAL continuation: the
                              initial-guess history of the MAC solve (two potentials per level + the dt they belong to), the step counter of
                              every level, the single-level driver's dt estimate, the box owners
  Level_<l>/MacPhiHist_<q>_*  the two MAC potentials

Old data are always written: the pressure of two steps ago seeds the initial guess of the next level projection (navierstokes.hip), so a
restart from new data alone would converge to the same answer along a different path -- equal to solver tolerance, not to the bit.
Host-side I/O (control plane): plain Python + numpy, one rank (the driver gathers nothing across ranks: multi-rank checkpoints are not
implemented and refused)."""
import ctypes as C
import json
import os
import numpy as np

from .plotfile import REAL_DESC, _fmt17

STATE_TYPES = [  # (name, selector new, selector old, index type, ncomp)
    ("State_Type", 0, 1, (0, 0, 0), 5),      # 5 + do_trac2 + do_temp components (the level's nstate)
    ("Press_Type", 2, 3, (1, 1, 1), 1),
    ("Gradp_Type", 4, 5, (0, 0, 0), 3),
]


def _box(lo, hi, typ=(0, 0, 0)):
    return "((" + ",".join(str(v) for v in lo) + ") (" + ",".join(str(v) for v in hi) + ") (" + ",".join(str(v) for v in typ) + "))"


def _write_vismf(dirname, name, boxes, typ, arrays, ngrow):
    """VisMF::Write of one MultiFab: <name>_H + <name>_D_00000; arrays: per box (nx + 2 ng, ..., ncomp) incl. ghost cells; boxes: the
    valid cell boxes (written converted to the index type, as BoxArray::writeOn does)"""
    nc = arrays[0].shape[-1] if arrays else 0
    fname = os.path.basename(name) + "_D_00000"
    offsets, mins, maxs = [], [], []
    with open(os.path.join(dirname, fname), "wb") as f:
        for (lo, hi), a in zip(boxes, arrays):
            a = np.asarray(a, dtype="<f8")
            glo = [lo[d] - ngrow for d in range(3)]
            ghi = [hi[d] + typ[d] + ngrow for d in range(3)]
            assert a.shape[:3] == tuple(ghi[d] - glo[d] + 1 for d in range(3)), (a.shape, glo, ghi)
            offsets.append(f.tell())
            f.write(f"FAB {REAL_DESC}{_box(glo, ghi, typ)} {nc}\n".encode())
            f.write(np.asfortranarray(a).tobytes(order="F"))
            mins.append([a[..., n].min() for n in range(nc)])
            maxs.append([a[..., n].max() for n in range(nc)])
    with open(os.path.join(dirname, os.path.basename(name) + "_H"), "w") as f:
        f.write(f"1\n0\n{nc}\n{ngrow}\n({len(boxes)} 0\n")
        for lo, hi in boxes:
            f.write(_box(lo, [hi[d] + typ[d] for d in range(3)], typ) + "\n")
        f.write(f")\n{len(boxes)}\n")
        for o in offsets:
            f.write(f"FabOnDisk: {fname} {o}\n")
        for vals in (mins, maxs):
            f.write(f"\n{len(boxes)},{nc}\n")
            for row in vals:
                f.write("".join(_fmt17(v) + "," for v in row) + "\n")


def _read_vismf(dirname, name):
    import re
    with open(os.path.join(dirname, name + "_H")) as f:
        txt = f.read()
    files = [(m.group(1), int(m.group(2))) for m in re.finditer(r"FabOnDisk: (\S+) (\d+)", txt)]
    out = []
    for fn, off in files:
        with open(os.path.join(dirname, fn), "rb") as f:
            f.seek(off)
            head = f.readline().decode()
            m = re.search(r"\(\(([-\d,]+)\) \(([-\d,]+)\) \(([-\d,]+)\)\) (\d+)\s*$", head)
            lo = [int(v) for v in m.group(1).split(",")]
            hi = [int(v) for v in m.group(2).split(",")]
            n = int(m.group(4))
            shape = tuple(h - q + 1 for q, h in zip(lo, hi)) + (n,)
            out.append(np.frombuffer(f.read(8 * int(np.prod(shape))), dtype="<f8").reshape(shape, order="F").astype(np.float64))
    return out


def _geom_line(g, n):
    # amrex::Geometry operator<<: coordinate system, the physical box, the index domain
    lo, hi = list(g.prob_lo), list(g.prob_hi)
    return "0 " + " ".join(repr(float(v)) for v in lo) + " " + " ".join(repr(float(v)) for v in hi) + " " + _box((0, 0, 0), [v - 1 for v in n])


def _level_views(run):
    """(levels, layouts, geoms, hierarchy or None) of an Amr hierarchy or a single NavierStokes level"""
    if hasattr(run, "levels"):
        return run.levels, run.layouts, [run.level_geom(l) for l in range(run.nlev)], run
    return [run], [run.layout], [run.geom], None


def write(run, root, step, max_level=None):
    """checkpoint of `run` (iamr_amd.amr.Amr or iamr_amd.ns.NavierStokes) into <root><step:05d>; returns the directory"""
    from .lib import lib, check
    L = lib()
    levels, lays, geoms, amr = _level_views(run)
    nlev = len(levels)
    path = f"{root}{step:05d}"
    os.makedirs(path, exist_ok=True)
    max_level = nlev - 1 if max_level is None else max_level
    dt_level, dt_min, n_cycle = (C.c_double * nlev)(), (C.c_double * nlev)(), (C.c_int * nlev)()
    counters, stop = (C.c_int * 2)(), C.c_double(-1.0)
    nmax = max(max_level, nlev - 1) + 1
    level_count = (C.c_int * nmax)()
    if amr is not None:
        check(L.iamrx_amr_restart_state(amr.h, 0, dt_level, dt_min, n_cycle, counters, C.byref(stop)))
        check(L.iamrx_amr_level_counts(amr.h, 0, level_count, nmax))
    states = []
    for lev in levels:
        st = (C.c_double * 16)()
        check(L.iamrx_ns_restart_state(lev.h, 0, st))
        states.append(list(st))
    if amr is None:
        dt_level[0], dt_min[0], n_cycle[0] = states[0][1], states[0][12], 1
        counters[0] = counters[1] = int(states[0][2])
        level_count[0] = counters[1]
        stop.value = states[0][13]
    H = ["CheckPointVersion_1.0", "3", repr(states[0][0]), str(max_level), str(nlev - 1)]
    for l in range(max_level + 1):
        H.append(_geom_line(geoms[min(l, nlev - 1)], [geoms[0].n[d] * 2 ** l for d in range(3)]))
    H.append(" ".join(["2"] * max_level))
    # Amr::checkPoint writes these five arrays with max_level + 1 entries; levels that do not exist (yet) carry what Amr gives a level
    # that is created by a later regrid: dt of the level below / 2, n_cycle 2, no steps
    pad = nmax - nlev
    dtl = [float(v) for v in dt_level] + [float(dt_level[nlev - 1]) / 2 ** (i + 1) for i in range(pad)]
    dtm = [float(v) for v in dt_min] + [float(dt_min[nlev - 1]) for i in range(pad)]
    H.append(" ".join(repr(v) for v in dtl))
    H.append(" ".join(repr(v) for v in dtm))
    H.append(" ".join([str(int(v)) for v in n_cycle] + ["2"] * pad))
    H.append(" ".join([str(int(s[2])) for s in states] + ["0"] * pad))           # level_steps
    H.append(" ".join(str(int(v)) for v in level_count))                         # level_count
    extra = {"stop_time": stop.value, "level_steps0": int(counters[0]), "level_count": int(counters[1]),
             "level_counts": [int(v) for v in level_count], "levels": []}
    for l, (lev, lay, g) in enumerate(zip(levels, lays, geoms)):
        ld = os.path.join(path, f"Level_{l}")
        os.makedirs(ld, exist_ok=True)
        boxes = [(list(lo), list(hi)) for lo, hi in lay.boxes]
        if any(o != 0 for o in lay.owners):
            raise NotImplementedError("checkpoint: multi-rank checkpoints are not implemented")
        n = [geoms[0].n[d] * 2 ** l for d in range(3)]
        H += [str(l), _geom_line(g, n), f"({len(boxes)} 0"] + [_box(lo, hi) for lo, hi in boxes] + [")", str(len(STATE_TYPES))]
        st = states[l]
        for t, (name, snew, sold, typ, nc) in enumerate(STATE_TYPES):
            dom_hi = [n[d] - 1 + typ[d] for d in range(3)]
            H += [_box((0, 0, 0), dom_hi, typ), f"({len(boxes)} 0"] + [_box(lo, [hi[d] + typ[d] for d in range(3)], typ) for lo, hi in boxes] + [")"]
            if t == 0:
                told, tnew = (st[4], st[4]), (st[3], st[3])          # Point type: start == stop
            else:
                told, tnew = (st[7], st[8]), (st[5], st[6])          # Interval types
            H += [repr(told[0]), repr(told[1]), repr(tnew[0]), repr(tnew[1]), "2", f"Level_{l}/SD_{t}_New_MF", f"Level_{l}/SD_{t}_Old_MF"]
            for tag, sel in (("New", snew), ("Old", sold)):
                mf = lev.data(sel)
                arrays = [mf.to_numpy(li)[0] for li in range(mf.nlocal())]
                _write_vismf(ld, f"SD_{t}_{tag}_MF", boxes, typ, arrays, 1)
        for q in range(2):
            mf = lev.data(10 + q)
            _write_vismf(ld, f"MacPhiHist_{q}", boxes, (0, 0, 0), [mf.to_numpy(li)[0] for li in range(mf.nlocal())], 0)
        extra["levels"].append({"state": st, "boxes": boxes, "owners": list(lay.owners)})
    with open(os.path.join(path, "Header"), "w") as f:
        f.write("\n".join(H) + "\n")
    with open(os.path.join(path, "iamrx_restart.json"), "w") as f:
        json.dump(extra, f)
    return path


def _types_patch():
    from .ns import NavierStokes
    NavierStokes._types.setdefault(10, ((0, 0, 0), 1, 0))
    NavierStokes._types.setdefault(11, ((0, 0, 0), 1, 0))


_types_patch()


def read_header(path):
    """the hierarchy part of a checkpoint Header: dict(time, max_level, finest_level, dt_level, dt_min, n_cycle, level_steps, level_count,
    boxes per level)"""
    import re
    with open(os.path.join(path, "Header")) as f:
        T = f.read().split("\n")
    assert T[0].strip() == "CheckPointVersion_1.0", T[0]
    time, max_level, finest = float(T[2]), int(T[3]), int(T[4])
    q = 5 + (max_level + 1) + 1
    dt_level = [float(v) for v in T[q].split()]
    dt_min = [float(v) for v in T[q + 1].split()]
    n_cycle = [int(v) for v in T[q + 2].split()]
    level_steps = [int(v) for v in T[q + 3].split()]
    level_count = [int(v) for v in T[q + 4].split()]
    q += 5
    boxes = []
    for l in range(finest + 1):
        assert int(T[q]) == l
        nb = int(T[q + 2].strip("(").split()[0])
        bl = []
        for b in range(nb):
            m = re.match(r"\(\(([-\d,]+)\) \(([-\d,]+)\)", T[q + 3 + b])
            bl.append((tuple(int(v) for v in m.group(1).split(",")), tuple(int(v) for v in m.group(2).split(","))))
        boxes.append(bl)
        q += 3 + nb + 1            # level, geom, "(n 0", boxes, ")"
        ntypes = int(T[q]); q += 1
        for _ in range(ntypes):
            q += 1                 # domain
            q += 1 + nb + 1        # BoxArray
            q += 4                 # times
            nmf = int(T[q]); q += 1 + nmf
    return dict(time=time, max_level=max_level, finest_level=finest, dt_level=dt_level, dt_min=dt_min, n_cycle=n_cycle, level_steps=level_steps,
                level_count=level_count, boxes=boxes)


def restart(path, geom0, params, opts=None, single_level=False, stop_time=None):
    """amr.restart: rebuild the run from a checkpoint directory.  geom0 / params / opts / stop_time come from the inputs file, as upstream
    re-reads them (a restart with a later stop_time is the usual reason to restart; the checkpoint's own stop_time is used only if none is
    given); returns an iamr_amd.amr.Amr (or a NavierStokes level if single_level and the checkpoint holds one level) that continues
    exactly where the checkpointed run stood -- no post_init."""
    from . import lib as Lb
    from .lib import lib, check
    from .ns import NavierStokes
    from .amr import Amr
    L = lib()
    hd = read_header(path)
    with open(os.path.join(path, "iamrx_restart.json")) as f:
        extra = json.load(f)
    nlev = hd["finest_level"] + 1
    lays = [Lb.Layout([(tuple(lo), tuple(hi)) for lo, hi in hd["boxes"][l]]) for l in range(nlev)]
    if single_level and nlev == 1:
        run = NavierStokes(geom0, lays[0], params, opts)
        levels = [run]
    else:
        run = Amr(geom0, lays, params, opts)
        levels = run.levels
    for l, lev in enumerate(levels):
        ld = os.path.join(path, f"Level_{l}")
        for t, (name, snew, sold, typ, nc) in enumerate(STATE_TYPES):
            for tag, sel in (("New", snew), ("Old", sold)):
                arrays = _read_vismf(ld, f"SD_{t}_{tag}_MF")
                mf = Lb.MultiFab(lays[l], typ, arrays[0].shape[-1] if (t == 0 and arrays) else nc, 1)
                for li, a in enumerate(arrays):
                    mf.from_numpy(a, li)
                lev.set_data(sel, mf)
        for q in range(2):
            arrays = _read_vismf(ld, f"MacPhiHist_{q}")
            mf = Lb.MultiFab(lays[l], (0, 0, 0), 1, 0)
            for li, a in enumerate(arrays):
                mf.from_numpy(a, li)
            lev.set_data(10 + q, mf)
        st = (C.c_double * 16)(*extra["levels"][l]["state"])
        if stop_time is not None:
            st[13] = float(stop_time)
        check(L.iamrx_ns_restart_state(lev.h, 1, st))
    if hasattr(run, "levels"):
        dt_level, dt_min = (C.c_double * nlev)(*hd["dt_level"][:nlev]), (C.c_double * nlev)(*hd["dt_min"][:nlev])
        n_cycle = (C.c_int * nlev)(*hd["n_cycle"][:nlev])
        counters = (C.c_int * 2)(extra["level_steps0"], extra["level_count"])
        stop = C.c_double(extra["stop_time"] if stop_time is None else float(stop_time))
        check(L.iamrx_amr_restart_state(run.h, 1, dt_level, dt_min, n_cycle, counters, C.byref(stop)))
        lc = extra.get("level_counts", hd["level_count"])
        check(L.iamrx_amr_level_counts(run.h, 1, (C.c_int * len(lc))(*lc), len(lc)))
    return run
