#!/bin/bash
# SQ / TCC counters of the cell-centred smoother kernels in the isolated sweep loop (tools/bench_rb.py): run on the GPU box from the repo root
root=$(pwd); out=$root/gpurun_out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
i=0
for c in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" \
         "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_LDS_IDX_ACTIVE" \
         "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1)); rm -rf /tmp/prbc_$i
    timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prbc_$i -- python $root/tools/bench_rb.py ${RB_N:-256} > $out/pmc_rb_$i.log 2>&1
    f=$(find /tmp/prbc_$i -name '*counter_collection.csv' | head -1)
    test -n "$f" && cp "$f" $out/pmc_rb_$i.csv
done
cd $root
python - <<'PY'
import csv, collections, glob, os
out = os.path.join(os.getcwd(), "gpurun_out")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/pmc_rb_*.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_abec_gsrb" not in k: continue
        short = k.split("(")[0].replace("iamrx::", "").replace("void ", "")
        agg[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmc_rb_report.txt", "w") as fo:
    for k, d in agg.items():
        fo.write(k + "\n")
        for c, v in sorted(d.items()):
            fo.write("   %-24s mean %.4g  (n=%d)\n" % (c, sum(v) / len(v), len(v)))
print(open(out + "/pmc_rb_report.txt").read())
PY
