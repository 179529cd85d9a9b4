#!/bin/bash
# A kernel source rebuilt ON THE GPU BOX under extra compiler flags (a -D experiment switch that compiles pieces out: wrong results, right
# costs; or back-end options such as -mllvm -amdgpu-sched-strategy=max-ilp), timed with a command, production object restored at the end.
#   bash tools/exp_build.sh k_nodal.hip "python tools/bench_gsr.py 256" "-DIAMRX_GSR_EXP=1" "-DIAMRX_GSR_EXP=4"
# Replaces the round-5 one-off scripts r5_gsr_exp.sh, r5_rbw_exp.sh, r5_interp_exp.sh, r5_god_flags.sh, r5_gsr_flags.sh, r5_god_ab.sh.
R=${GRAFT_REPO_ROOT:-$(pwd)}
src=$1; cmd=$2; shift 2
cd $R/iamr_amd/csrc
obj=${src%.hip}.o
fp="-ffp-contract=off"; [ "$src" = "k_godunov.hip" ] && fp="-ffp-contract=fast"
BASE="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-fast-math -Wall -Wno-unused-result $fp"
cp $obj /tmp/exp_base.o
link() { make -s 2>&1 | tail -1; }
echo "== production"; (cd $R && eval "$cmd")
for f in "$@"; do
    echo "== $f"
    if timeout 900 /opt/rocm/bin/hipcc $BASE $f -c $src -o $obj 2> /tmp/exp_build.err; then touch $obj; link; (cd $R && eval "$cmd"); else tail -3 /tmp/exp_build.err; fi
done
cp /tmp/exp_base.o $obj; touch $obj; link
