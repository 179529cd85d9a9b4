"""Kernel sequence between two kernels of a rocprofv3 kernel trace (CSV): python tools/trace_window.py trace.csv START END [occurrence]
Prints, for the chosen occurrence of a kernel whose name contains START up to the next kernel whose name contains END, the launches
grouped into runs of the same (name, grid) with their count and total time, plus the gaps between kernels."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
start, end = sys.argv[2], sys.argv[3]
occ = int(sys.argv[4]) if len(sys.argv) > 4 else -1
idx = [i for i, r in enumerate(rows) if start in r["Kernel_Name"]]
i0 = idx[occ]
i1 = next(i for i in range(i0 + 1, len(rows)) if end in rows[i]["Kernel_Name"])
t0 = int(rows[i0]["Start_Timestamp"])
runs = []
busy = 0
for r in rows[i0:i1 + 1]:
    name = r["Kernel_Name"].replace("iamrx::", "")[:70]
    key = (name, r["Grid_Size_X"])
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    busy += d
    if runs and runs[-1][0] == key:
        runs[-1][1] += 1; runs[-1][2] += d
    else:
        runs.append([key, 1, d, (int(r["Start_Timestamp"]) - t0) / 1e3])
span = (int(rows[i1]["End_Timestamp"]) - t0) / 1e3
print(f"window {span:.1f} us, kernels busy {busy / 1e3:.1f} us, launches {i1 - i0 + 1}")
for key, n, d, at in runs:
    print(f"{at:10.1f} us  {n:3d} x {d / n / 1e3:8.1f} us  {key[0]} g={key[1]}")
