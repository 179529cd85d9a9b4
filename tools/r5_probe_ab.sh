#!/bin/bash
# the bench's timed region with and without its in-step HIP-event probes, and tools/run_steps.py, on one lease
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
F="--steps 10 --warmup 2 --repeats 3 --no-cpu-baseline --no-multibox --no-shard-proxy --no-upstream-shape --amr-steps 0 --ldc-steps 0"
for p in 1 0 1 0; do
IAMRX_BENCH_PROBE=$p python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('probe=$p', d['ms_per_step'], d['ms_per_step_of_each_region'], d['host_syncs_per_step'])"
done
python tools/run_steps.py | grep ms/step
python tools/run_steps.py | grep ms/step
