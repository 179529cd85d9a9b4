#!/bin/bash
# rocprofv3 kernel trace + stats of the default bench run; per-grid summary of the nodal smoother.  Run on the GPU box from the repo root:
#   bash tools/profile_bench.sh TAG   -> gpurun_out/TAG_kernel_stats.csv, TAG_kernels_by_grid.csv, TAG_bench.json
tag=${1:-prof}
root=$(pwd)
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python $root/bench.py --no-cpu-baseline --rt-n 0 > $out/${tag}_bench.json 2> $out/${tag}_bench.err
f=$(find /tmp/prof_$tag -name '*kernel_stats.csv' | head -1)
test -n "$f" && cp "$f" $out/${tag}_kernel_stats.csv
t=$(find /tmp/prof_$tag -name '*kernel_trace.csv' | head -1)
test -n "$t" && python $root/tools/trace_by_grid.py "$t" $out/${tag}_kernels_by_grid.csv k_nodal_gsr k_nodal_gs4 k_abec_gsrb k_god_z k_pred_z
head -12 $out/${tag}_kernel_stats.csv | cut -c1-60,200-
head -6 $out/${tag}_kernels_by_grid.csv
