"""Where do the launches of one kernel sit in a step?  python tools/trace_context.py trace.csv START END PATTERN
Within the last window [kernel containing START ... next kernel containing END] of a rocprofv3 kernel trace, prints every launch whose
name contains PATTERN with its duration and the two kernels in front of and behind it."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
start, end, pat = sys.argv[2], sys.argv[3], sys.argv[4]
idx = [i for i, r in enumerate(rows) if start in r["Kernel_Name"]]
i0 = idx[-2] if len(idx) > 1 else idx[-1]
i1 = next(i for i in range(i0 + 1, len(rows)) if end in rows[i]["Kernel_Name"])
short = lambda r: r["Kernel_Name"].replace("iamrx::", "").replace("void ", "")[:48] + " g=" + r["Grid_Size_X"]
for i in range(i0, i1 + 1):
    r = rows[i]
    if pat in r["Kernel_Name"]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print(f"{d:8.1f} us g={r['Grid_Size_X']:>8}  <- {short(rows[i - 2])} | {short(rows[i - 1])}  -> {short(rows[i + 1])} | {short(rows[i + 2])}")
