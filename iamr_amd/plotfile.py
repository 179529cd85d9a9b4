"""AMReX-format plotfiles: writer, reader and an fcompare-style comparison.

Format restated from the reference's committed plotfile `Exec/run2d/test_grids/plt0000_1` (Header, Level_*/Cell_H, Cell_D_* --
kept as a data fixture under tests/golden/plt0000_1) and from the call sites that produce it (NavierStokesBase::writePlotFile,
Source/NavierStokesBase.cpp:3344-3352 / AmrLevel::writePlotFile role):

  <dir>/Header            version string, ncomp, names, dim, time, finest_level, prob_lo, prob_hi, ref_ratio, domains, level steps,
                          dx per level, coord_sys, 0, then per level: "lev ngrids time", step, the physical extent of every grid,
                          and the relative path "Level_<l>/Cell"
  <dir>/Level_<l>/Cell_H  VisMF header: version 1, how 0, ncomp, ngrow, the BoxArray, "FabOnDisk: file offset" per grid, and the
                          per-grid / per-component minima and maxima
  <dir>/Level_<l>/Cell_D_00000   per grid: "FAB ((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))(box) ncomp\\n" + the doubles
                          (little endian, Fortran order, one component after the other)

Host-side I/O (control plane): plain Python + numpy, dimension-generic (2-D files of the reference can be read)."""
import os
import re
import numpy as np

REAL_DESC = "((8, (64 11 52 0 1 12 0 1023)),(8, (8 7 6 5 4 3 2 1)))"


def _fmt(x):
    """shortest round-trip representation, the way operator<< prints the reference's numbers (0.0625, 0, 1, 0.5 ...)"""
    x = float(x)
    if x == int(x) and abs(x) < 1e15:
        return str(int(x))
    return repr(x)


def _fmt17(x):
    return "%.15g" % x if float("%.15g" % x) == x else "%.17g" % x


def _box_str(lo, hi):
    dim = len(lo)
    return "((" + ",".join(str(v) for v in lo) + ") (" + ",".join(str(v) for v in hi) + ") (" + ",".join("0" for _ in range(dim)) + "))"


class Level:
    """one AMR level of a plotfile: index domain, mesh spacing, boxes and the data of every box (array (n..., ncomp), Fortran order)"""

    def __init__(self, domain, dx, boxes, data=None, step=0, time=0.0):
        self.domain = (tuple(domain[0]), tuple(domain[1]))
        self.dx = tuple(dx)
        self.boxes = [(tuple(lo), tuple(hi)) for lo, hi in boxes]
        self.data = data
        self.step = step
        self.time = time
        self.fab_files = None      # reader: (file, offset) per box


class PlotFile:
    def __init__(self, names, time, prob_lo, prob_hi, levels, ref_ratio=None, coord_sys=0, version="HyperCLaw-V1.1"):
        self.names = list(names)
        self.time = time
        self.prob_lo = tuple(prob_lo)
        self.prob_hi = tuple(prob_hi)
        self.levels = levels
        self.ref_ratio = list(ref_ratio) if ref_ratio is not None else [2] * (len(levels) - 1)
        self.coord_sys = coord_sys
        self.version = version

    # ------------------------------------------------------------------------------------------------ writer
    def header_text(self):
        dim = len(self.prob_lo)
        L = [self.version, str(len(self.names))] + self.names + [str(dim), _fmt(self.time), str(len(self.levels) - 1)]
        L.append(" ".join(_fmt(v) for v in self.prob_lo) + " ")
        L.append(" ".join(_fmt(v) for v in self.prob_hi) + " ")
        L.append(" ".join(str(r) for r in self.ref_ratio) + (" " if self.ref_ratio else ""))
        L.append(" ".join(_box_str(*lv.domain) for lv in self.levels) + " ")
        L.append(" ".join(str(lv.step) for lv in self.levels) + " ")
        for lv in self.levels:
            L.append(" ".join(_fmt(v) for v in lv.dx) + " ")
        L += [str(self.coord_sys), "0"]
        for l, lv in enumerate(self.levels):
            L.append(f"{l} {len(lv.boxes)} {_fmt(lv.time)}")
            L.append(str(lv.step))
            for lo, hi in lv.boxes:
                for d in range(dim):
                    L.append(f"{_fmt(self.prob_lo[d] + lo[d] * lv.dx[d])} {_fmt(self.prob_lo[d] + (hi[d] + 1) * lv.dx[d])}")
            L.append(f"Level_{l}/Cell")
        return "\n".join(L) + "\n"

    def write(self, path):
        os.makedirs(path, exist_ok=True)
        with open(os.path.join(path, "Header"), "w") as f:
            f.write(self.header_text())
        nc = len(self.names)
        for l, lv in enumerate(self.levels):
            ld = os.path.join(path, f"Level_{l}")
            os.makedirs(ld, exist_ok=True)
            offsets, mins, maxs = [], [], []
            fname = "Cell_D_00000"
            with open(os.path.join(ld, fname), "wb") as f:
                for (lo, hi), a in zip(lv.boxes, lv.data):
                    a = np.asarray(a, dtype="<f8")
                    assert a.shape == tuple(h - q + 1 for q, h in zip(lo, hi)) + (nc,), (a.shape, lo, hi)
                    offsets.append(f.tell())
                    f.write(f"FAB {REAL_DESC}{_box_str(lo, hi)} {nc}\n".encode())
                    f.write(np.asfortranarray(a).tobytes(order="F"))
                    mins.append([a[..., n].min() for n in range(nc)])
                    maxs.append([a[..., n].max() for n in range(nc)])
            with open(os.path.join(ld, "Cell_H"), "w") as f:
                f.write(f"1\n0\n{nc}\n0\n({len(lv.boxes)} 0\n")
                for lo, hi in lv.boxes:
                    f.write(_box_str(lo, hi) + "\n")
                f.write(f")\n{len(lv.boxes)}\n")
                for o in offsets:
                    f.write(f"FabOnDisk: {fname} {o}\n")
                for vals in (mins, maxs):
                    f.write(f"\n{len(lv.boxes)},{nc}\n")
                    for row in vals:
                        f.write("".join(_fmt17(v) + "," for v in row) + "\n")

    # ------------------------------------------------------------------------------------------------ reader
    @staticmethod
    def read(path, load_data=True):
        with open(os.path.join(path, "Header")) as f:
            T = f.read().split("\n")
        it = iter(T)
        version = next(it).strip()
        nc = int(next(it))
        names = [next(it).strip() for _ in range(nc)]
        dim = int(next(it))
        time = float(next(it))
        finest = int(next(it))
        prob_lo = tuple(float(v) for v in next(it).split())
        prob_hi = tuple(float(v) for v in next(it).split())
        ref_ratio = [int(v) for v in next(it).split()]
        doms = re.findall(r"\(\(([-\d,]+)\) \(([-\d,]+)\) \([-\d,]+\)\)", next(it))
        steps = [int(v) for v in next(it).split()]
        dxs = [tuple(float(v) for v in next(it).split()) for _ in range(finest + 1)]
        coord = int(next(it))
        next(it)
        levels = []
        for l in range(finest + 1):
            lev, ngrids, ltime = next(it).split()
            lstep = int(next(it))
            for _ in range(int(ngrids) * dim):
                next(it)
            rel = next(it).strip()
            dlo, dhi = (tuple(int(v) for v in s.split(",")) for s in doms[l])
            lv = Level((dlo, dhi), dxs[l], [], None, lstep, float(ltime))
            PlotFile._read_level(path, rel, lv, nc, load_data)
            levels.append(lv)
        return PlotFile(names, time, prob_lo, prob_hi, levels, ref_ratio, coord, version)

    @staticmethod
    def _read_level(path, rel, lv, nc, load_data):
        with open(os.path.join(path, rel + "_H")) as f:
            txt = f.read()
        boxes = re.findall(r"\(\(([-\d,]+)\) \(([-\d,]+)\) \([-\d,]+\)\)", txt)
        lv.boxes = [(tuple(int(v) for v in lo.split(",")), tuple(int(v) for v in hi.split(","))) for lo, hi in boxes]
        lv.fab_files = [(m.group(1), int(m.group(2))) for m in re.finditer(r"FabOnDisk: (\S+) (\d+)", txt)]
        assert len(lv.fab_files) == len(lv.boxes)
        if not load_data:
            return
        lv.data = []
        ld = os.path.dirname(os.path.join(path, rel))
        for (lo, hi), (fn, off) in zip(lv.boxes, lv.fab_files):
            with open(os.path.join(ld, fn), "rb") as f:
                f.seek(off)
                head = f.readline().decode()
                m = re.match(r"FAB \(\((\d+), \(([\d ]+)\)\),\((\d+), \(([\d ]+)\)\)\)", head)
                assert m and int(m.group(1)) == 8, head
                order = [int(v) for v in m.group(4).split()]
                little = order == [8, 7, 6, 5, 4, 3, 2, 1]
                assert little or order == [1, 2, 3, 4, 5, 6, 7, 8], head
                n = int(head.rsplit(" ", 1)[1])
                shape = tuple(h - q + 1 for q, h in zip(lo, hi)) + (n,)
                cnt = int(np.prod(shape))
                a = np.frombuffer(f.read(8 * cnt), dtype="<f8" if little else ">f8").reshape(shape, order="F")
                lv.data.append(a.astype(np.float64))


def compare(path_a, path_b):
    """fcompare role: {name: (abs Linf, rel Linf)} of the level-wise difference of two plotfiles with identical grids"""
    A, B = PlotFile.read(path_a), PlotFile.read(path_b)
    if A.names != B.names or len(A.levels) != len(B.levels):
        raise ValueError("plotfiles differ in variables or number of levels")
    out = {}
    for n, name in enumerate(A.names):
        ea, ma = 0.0, 0.0
        for la, lb in zip(A.levels, B.levels):
            if la.boxes != lb.boxes:
                raise ValueError("plotfiles differ in their grids")
            for a, b in zip(la.data, lb.data):
                ea = max(ea, float(np.abs(a[..., n] - b[..., n]).max()))
                ma = max(ma, float(np.abs(a[..., n]).max()))
        out[name] = (ea, ea / ma if ma > 0 else 0.0)
    return out


STATE_NAMES_3D = ["x_velocity", "y_velocity", "z_velocity", "density", "tracer"]


def state_names(do_trac2=0, do_temp=0):
    """names of the State_Type components (NS_setup.cpp:250-283), then divu and dsdt (Divu_Type, Dsdt_Type) in a temperature run"""
    return STATE_NAMES_3D + (["tracer2"] if do_trac2 else []) + (["temp", "divu", "dsdt"] if do_temp else [])


DERIVE_NAMES = ["energy", "mag_vort", "avg_pressure"]          # derive_lst order (NS_setup.cpp:436-449; no particles, no time averages)


def plot_selection(state, plot_vars="ALL", derive_plot_vars="NONE"):
    """(indices of the state components, names of the derived quantities) a plotfile holds: amr.plot_vars picks state variables (ALL: every
    one), amr.derive_plot_vars derived ones (ALL: the derive list in its order; default NONE) -- Amr::initPltAndChk / fillDerivePlotVarList.
    Unknown names raise, as amrex::Amr aborts on them."""
    if plot_vars == "ALL":
        keep = list(range(len(state)))
    elif plot_vars == "NONE":
        keep = []
    else:
        bad = [v for v in plot_vars if v not in state]
        if bad:
            raise ValueError(f"amr.plot_vars: not state variables: {bad} (have {state})")
        keep = [q for q, nm in enumerate(state) if nm in plot_vars]           # plotfile order = state order (Amr::statePlotVars is a list filled in descriptor order)
    if derive_plot_vars == "ALL":
        der = list(DERIVE_NAMES)
    elif derive_plot_vars == "NONE":
        der = []
    else:
        bad = [v for v in derive_plot_vars if v not in DERIVE_NAMES]
        if bad:
            raise ValueError(f"amr.derive_plot_vars: unknown derived quantities {bad} (have {DERIVE_NAMES})")
        der = list(derive_plot_vars)
    return keep, der


def from_level_data(geom_n, prob_lo, prob_hi, boxes, arrays, time, step, names=None):
    """single-level PlotFile from per-box arrays (valid region, (nx,ny,nz,ncomp))"""
    dim = len(geom_n)
    dx = [(prob_hi[d] - prob_lo[d]) / geom_n[d] for d in range(dim)]
    lv = Level(((0,) * dim, tuple(v - 1 for v in geom_n)), dx, boxes, arrays, step, time)
    return PlotFile(names or STATE_NAMES_3D, time, prob_lo, prob_hi, [lv])
