#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "1 1" "0 1" "1 0" "0 0"; do
set -- $v
rm -rf /tmp/pn
IAMRX_NODAL_RES_XCD=$1 IAMRX_NODAL_RES_FAT=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pn -- python $R/tools/bench_nodal_ops.py 256 > /tmp/o.log 2>&1
f=$(find /tmp/pn -name '*kernel_stats.csv' | head -1)
echo "XCD=$1 FAT=$2"; grep "nodal_res" $f | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin): print('   ',r[0][:50], r[1], round(float(r[3])/1e3,1))"
done
