#!/bin/bash
# HBM traffic and vector-instruction counts of the kernels inside the running bench step: three rocprofv3 counter passes (FETCH_SIZE,
# WRITE_SIZE cannot share a pass; SQ_INSTS_VALU for the valu_frac of the Godunov kernels,
# MI355X_MICROARCH.md "rocprofv3 PMC slots") over bench.py, --kernel-trace only.  Run on the GPU box from the repo root:
#   bash tools/collect_pmc.sh roundN      -> gpurun_out/roundN_pmc.json (+ the two reduced CSVs); copy them to profiles/
set -e
tag=${1:-round}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    rm -rf /tmp/pmc_$c
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -- python $root/bench.py --steps 2 --warmup 1 --repeats 1 --no-cpu-baseline --no-multibox --no-shard-proxy --no-upstream-shape --amr-steps 0 --ldc-steps 0 --c3-n 0 --rt-n 0 > $out/${tag}_pmc_$c.log 2>&1
    f=$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)
    test -n "$f" && cp "$f" $out/${tag}_pmc_$(echo $c | tr A-Z a-z).csv
done
cd $root
python tools/pmc_report.py $out/${tag}_pmc_fetch_size.csv $out/${tag}_pmc_write_size.csv $out/${tag}_pmc.json $out/${tag}_pmc_sq_insts_valu.csv > $out/${tag}_pmc_report.txt
python - <<PY
import json, subprocess, sys
sys.path.insert(0, "$root")
import bench
p = "$out/${tag}_pmc.json"
d = json.load(open(p))
d["source_blobs"] = {f: bench.file_blob_sha("$root/iamr_amd/csrc/" + f) for f in ("k_nodal.hip", "k_abec.hip", "k_godunov.hip")}
d["note"] = d["note"].replace("tools/pmc_kernels.py 256", "python bench.py --steps 2 --warmup 1 --cpu-steps 0 --amr-steps 0 (kernels INSIDE the running step, keyed by kernel and launch grid)")
json.dump(d, open(p, "w"), indent=1)
PY
# the raw counter CSVs (tens of MB each) stay on the box: gpurun merges at most 64 MiB back
rm -f $out/${tag}_pmc_fetch_size.csv $out/${tag}_pmc_write_size.csv $out/${tag}_pmc_sq_insts_valu.csv
head -30 $out/${tag}_pmc_report.txt
