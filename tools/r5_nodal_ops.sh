#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R; timeout 600 python -m pytest tests/test_gpu_ns.py tests/test_gpu_nodal_dirichlet.py tests/test_gpu_nodal_gsr.py tests/test_gpu_nodal_fused.py tests/test_gpu_walls.py -q -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pn
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pn -- python $R/tools/bench_nodal_ops.py ${1:-256} 2>&1 | grep "tile"
f=$(find /tmp/pn -name '*kernel_stats.csv' | head -1)
grep "nodal" $f | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin): print(r[0][:80], r[1], round(float(r[3])/1e3,1))"
