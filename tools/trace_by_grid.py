"""Per-grid-size summary of a rocprofv3 kernel trace (the --stats table averages a kernel over all multigrid levels):
   python tools/trace_by_grid.py KERNEL_TRACE.csv OUT.csv [name filter ...]"""
import collections
import csv
import re
import sys


def main():
    src, out = sys.argv[1:3]
    filt = sys.argv[3:]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(src)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        if filt and not any(f in name for f in filt):
            continue
        acc[(name, int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r.get("Grid_Size_Z", 1) or 1))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs"])
        for (name, gx, gy, gz), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):      # gz: components of the fused Godunov launches
            w.writerow([name, gx, gy, gz, len(v), sum(v), round(sum(v) / len(v), 1), min(v), max(v)])


if __name__ == "__main__":
    main()
