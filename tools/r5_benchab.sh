#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
A="--no-cpu-baseline --no-multibox --no-shard-proxy --no-upstream-shape --amr-steps 0 --ldc-steps 0 --c3-n 0"
python tools/run_steps.py 2>&1 | grep "ms/step"
python bench.py $A 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['ms_per_step_of_each_region'], d['mlmg_vcycle_ms'])"
IAMRX_BENCH_PROBE=0 python bench.py $A 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench noprobe', d['ms_per_step'], d['ms_per_step_of_each_region'], d['mlmg_vcycle_ms'])"
python tools/run_steps.py 2>&1 | grep "ms/step"
