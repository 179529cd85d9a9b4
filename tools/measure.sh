#!/bin/bash
# The measurement set of a round: the PMC passes (the bench line's `traffic` comes from them), rocprofv3 kernel statistics of the bench command,
# the plain bench line, per-step kernel tables of the three workloads, host-side HIP statistics, the C5 kernel statistics.
#   bash tools/measure.sh round6_final      -> gpurun_out/<tag>_*, profiles/<round>_pmc.json       (round 5: tools/r5_final.sh + r5_state.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-round6_final}
round=${tag%%_*}
out=$R/gpurun_out; mkdir -p $out
cd $R
if [ -z "$SKIP_PMC" ]; then
    bash tools/collect_pmc.sh $round > $out/${tag}_pmc.log 2>&1
    tail -3 $out/${tag}_pmc.log
    cp $out/${round}_pmc.json $R/profiles/${round}_pmc.json
fi
bash tools/profile_bench.sh $tag > $out/${tag}_profile.log 2>&1
tail -12 $out/${tag}_profile.log
# (profile_bench.sh leaves the bench line of the run UNDER rocprofv3 in ${tag}_bench.json: kept under its own name; the plain run is the record)
mv $out/${tag}_bench.json $out/${tag}_bench_under_rocprofv3.json
timeout 1500 python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
tail -c 300 $out/${tag}_bench.json
export NAMELEN=90 NTOP=45
bash tools/profile_step.sh > $out/${round}_step_1box.txt 2>&1
DBG=run_ldc_steps.py bash tools/profile_step.sh > $out/${round}_step_ldc256.txt 2>&1
AMR_N0=256 DBG=run_amr_steps.py bash tools/profile_step.sh > $out/${round}_step_amr256.txt 2>&1
grep -h "ms/step\|kernel ms" $out/${round}_step_*.txt
bash tools/hip_host.sh > $out/${tag}_hip_host.txt 2>&1
DBG=run_ldc_steps.py bash tools/hip_host.sh >> $out/${tag}_hip_host.txt 2>&1
grep "cadence\|ms/step" $out/${tag}_hip_host.txt
[ -n "$SKIP_RT" ] || TAG=${round}_rt bash tools/prof_rt.sh > $out/${tag}_rt.log 2>&1
