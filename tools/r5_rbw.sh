#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for nw in 0 1; do
rm -rf /tmp/prb
IAMRX_GSRB_RB_NW12=$nw RB_PER=0,0,0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prb -- python $R/tools/bench_rb.py 256 > /tmp/rb.log 2>&1
f=$(find /tmp/prb -name '*kernel_stats.csv' | head -1)
echo "NW12=$nw"; grep "gsrb_rb" $f | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin): print(r[0][:70], r[1], round(float(r[3])/1e3,1))"
done
cd $R; timeout 600 python -m pytest tests/test_gpu_kernel_forms.py tests/test_gpu_walls.py tests/test_gpu_ldc.py -q -x 2>&1 | tail -3
