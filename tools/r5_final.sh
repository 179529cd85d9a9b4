#!/bin/bash
# round-5 measurement set: the default bench line, rocprofv3 kernel statistics of the same command, the PMC passes, per-step kernel tables
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-round5_b}
out=$R/gpurun_out; mkdir -p $out
cd $R
timeout 900 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -c 600 $out/${tag}_bench.json
bash tools/collect_pmc.sh round5 > $out/${tag}_pmc.log 2>&1
tail -5 $out/${tag}_pmc.log
bash tools/profile_bench.sh $tag > $out/${tag}_profile.log 2>&1
tail -20 $out/${tag}_profile.log
bash tools/r5_state.sh
