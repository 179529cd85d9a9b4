#!/bin/bash
# SQ / TCC counters of the kernels whose names match $PAT in `python tools/$SCRIPT $ARGS`: run on the GPU box from the repo root
#   PAT=k_nodal_res SCRIPT=bench_nodal_ops.py bash tools/pmc_any.sh   -> gpurun_out/pmc_any_report.txt
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); rm -rf /tmp/pa_$i
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pa_$i -- python $root/tools/$SCRIPT $ARGS > $out/pmc_any_$i.log 2>&1
    f=$(find /tmp/pa_$i -name '*counter_collection.csv' | head -1)
    test -n "$f" && cp "$f" $out/pmc_any_$i.csv
done
cd $root
python - <<'PY'
import csv, collections, glob, os
out = os.path.join(os.getcwd(), "gpurun_out")
pat = os.environ.get("PAT", "k_").split(",")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/pmc_any_[0-9]*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not any(p in k for p in pat): continue
        short = k.split("(")[0].replace("iamrx::", "").replace("void ", "") + " g=" + r["Grid_Size"]
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_any_report.txt", "w") as fo:
    for k, d in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write("   %-24s mean %.4g  (n=%d)\n" % (c, sum(v) / len(v), len(v)))
print(open(out + "/pmc_any_report.txt").read())
PY
