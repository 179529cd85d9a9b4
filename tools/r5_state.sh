#!/bin/bash
# round-5 state: per-step kernel tables of the uniform 256^3 step, LidDrivenCavity 256^3 and the 256^3 + 256^3 AMR step
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out; mkdir -p $out
export NAMELEN=90 NTOP=45
cd $R
bash tools/profile_step.sh > $out/r5s_step_1box.txt 2>&1
DBG=run_ldc_steps.py bash tools/profile_step.sh > $out/r5s_step_ldc256.txt 2>&1
AMR_N0=256 DBG=run_amr_steps.py bash tools/profile_step.sh > $out/r5s_step_amr256.txt 2>&1
grep -h "ms/step\|kernel ms" $out/r5s_step_*.txt
