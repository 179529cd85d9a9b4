#!/bin/bash
# A/B of ONE run-time switch on one lease: the three workloads (uniform TaylorGreen 256^3, LidDrivenCavity 256^3, 2-level AMR 256^3 + 256^3)
# with IAMRX_<KEY> = each of the given values.   bash tools/ab_switch.sh KEY VALUE [VALUE ...]      (DESIGN.md section 9 lists the keys)
# Replaces the round-5 one-off scripts r5_acc.sh, r5_cf.sh, r5_nbr.sh, r5_refl.sh, r5_tail.sh, r5_graph.sh, r5_tb.sh, r5_res_ab.sh ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out; mkdir -p $out
cd $R
key=$1; shift
for v in "$@"; do
    export IAMRX_$key=$v
    echo "TG  $key=$v $(timeout 600 python tools/run_steps.py 2>&1 | grep 'ms/step')"
    [ -n "$SKIP_LDC" ] || echo "LDC $key=$v $(timeout 600 python tools/run_ldc_steps.py 2>&1 | grep 'ms/step')"
    [ -n "$SKIP_AMR" ] || timeout 600 python tools/bench_amr.py 256 3 2> $out/ab_${key}_$v.err | python3 -c "
import json,sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('AMR $key=$v', d['ms_per_coarse_step'], d['cells_advanced_per_sec'])"
    unset IAMRX_$key
done
