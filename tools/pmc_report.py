"""Turns the two rocprofv3 counter CSVs of tools/pmc_kernels.py (one --pmc FETCH_SIZE pass, one --pmc WRITE_SIZE pass) into
HBM bytes per launch per kernel:  python tools/pmc_report.py FETCH.csv WRITE.csv OUT.json
Unit and gfx950 correction as calibrated in profiles/round1_pmc.json (counter unit KB; FETCH_SIZE counts half of the bytes)."""
import csv, json, re, sys
from collections import defaultdict


def load(path, name):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "") + " grid=" + r["Grid_Size"]
        acc[k][0] += 1
        acc[k][1] += float(r["Counter_Value"])
    return acc


def main():
    f, w, out = sys.argv[1:4]
    F, W = load(f, "FETCH_SIZE"), load(w, "WRITE_SIZE")
    V = load(sys.argv[4], "SQ_INSTS_VALU") if len(sys.argv) > 4 else {}      # optional third pass: wave-level vector instructions
    kernels = {}
    for k in F:
        nf, sf = F[k]
        nw, sw = W.get(k, [0, 0.0])
        fe, wr = sf / max(nf, 1), sw / max(nw, 1)
        kernels[k] = {"launches": nf, "FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "hbm_bytes_per_launch": 2 * fe * 1024 + wr * 1024}
        if k in V:
            kernels[k]["SQ_INSTS_VALU"] = V[k][1] / max(V[k][0], 1)
    note = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of tools/pmc_kernels.py 256 on MI355X; "
            "counter unit KB. Calibration (k_fill / MultiFab::Copy of one 256^3 fp64 array = 134217728 B): WRITE_SIZE exact, FETCH_SIZE = 1/2 "
            "of the bytes (gfx950 correction of MI355X_MICROARCH.md) -> hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024.")
    json.dump({"note": note, "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in kernels.items():
        print(f"{k[:90]:90s} {v['launches']:4d} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB")


if __name__ == "__main__":
    main()
