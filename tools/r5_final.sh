#!/bin/bash
# round-5 measurement set: the PMC passes (the bench line's `traffic` comes from them), the default bench line, rocprofv3 kernel statistics of
# the same command, per-step kernel tables, host-side HIP statistics.  bash tools/r5_final.sh TAG -> gpurun_out/TAG_*, round5_pmc.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-round5_final}
out=$R/gpurun_out; mkdir -p $out
cd $R
bash tools/collect_pmc.sh round5 > $out/${tag}_pmc.log 2>&1
tail -3 $out/${tag}_pmc.log
cp $out/round5_pmc.json $R/profiles/round5_pmc.json
bash tools/profile_bench.sh $tag > $out/${tag}_profile.log 2>&1
tail -12 $out/${tag}_profile.log
# (profile_bench.sh leaves the bench line of the run UNDER rocprofv3 in ${tag}_bench.json: kept under its own name; the plain run is the record)
mv $out/${tag}_bench.json $out/${tag}_bench_under_rocprofv3.json
timeout 900 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -c 300 $out/${tag}_bench.json
bash tools/r5_state.sh
bash tools/r5_hip.sh > $out/${tag}_hip_host.txt 2>&1
DBG=run_ldc_steps.py bash tools/r5_hip.sh >> $out/${tag}_hip_host.txt 2>&1
cat $out/${tag}_hip_host.txt | grep "cadence\|ms/step"
