#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/iamr_amd/csrc
for e in ${EXPS:-0} 0; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -DIAMRX_INTERP_EXP=$e -c k_nodal.hip -o k_nodal.o 2>/dev/null && make -s 2>&1 | tail -1
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pg
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pg -- python $R/tools/bench_nodal_ops.py 256 > /tmp/g.log 2>&1
  f=$(find /tmp/pg -name '*kernel_stats.csv' | head -1)
  echo "EXP=$e"; grep "k_nodal_interp" $f | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin): print('  ', r[0][:60], r[1], round(float(r[3])/1e3,1))"
  cd $R/iamr_amd/csrc
done
