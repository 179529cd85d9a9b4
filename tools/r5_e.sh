#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_rb_nbr.py tests/test_gpu_dist.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3
for nw in 1 0; do
IAMRX_GSRB_RB_NW12=$nw python bench.py --steps 3 --warmup 1 --repeats 1 --no-multibox --no-upstream-shape --amr-steps 0 --ldc-steps 2 --c3-n 0 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); sp=d['shard_proxy']; print('NW12=$nw', {k:(round(v['ms_per_box_step'],2), [round(x,2) for x in v['mlmg_vcycle_ms']]) for k,v in sp.items() if isinstance(v,dict)}, 'ldc', round(d['lid_driven_cavity']['ms_per_step'],2), d['lid_driven_cavity']['mlmg_vcycle_ms'])"
done
