#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export NAMELEN=60 NTOP=200
for f in 1 0; do echo "MG_TAIL_FUSED=$f"; IAMRX_MG_TAIL_FUSED=$f bash tools/profile_step.sh > /tmp/t.txt 2>&1; head -3 /tmp/t.txt; grep "k_abec_tail\|k_abec_bottom\|gsrb1<0, 1, false> g=4096\|gsrb1<0, 1, false> g=16384" /tmp/t.txt; done
