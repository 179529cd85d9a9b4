#!/bin/bash
# timing experiments on the wall variant of the sweep kernel: rebuild k_abec.hip with pieces of the wall handling removed (IAMRX_RBW_EXP bits:
# wrong results) -- EXPS="0 1 2 4 8 15" bash tools/r5_rbw_exp.sh; the library is rebuilt in its production form (0) at the end
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/iamr_amd/csrc
for e in ${EXPS:-0} 0; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math $XFLAGS -DIAMRX_RBW_EXP=$e -c k_abec.hip -o k_abec.o 2>/dev/null && make -s 2>&1 | tail -1
  cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prb
  RB_PER=0,0,0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prb -- python $R/tools/bench_rb.py 256 > /tmp/rb.log 2>&1
  f=$(find /tmp/prb -name '*kernel_stats.csv' | head -1)
  echo "EXP=$e"; grep "gsrb_rb" $f | python3 -c "
import csv,sys
for r in csv.reader(sys.stdin): print('  ', r[0][:66], r[1], round(float(r[3])/1e3,1))"
  cd $R/iamr_amd/csrc
done
cd $R
