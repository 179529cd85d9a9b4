#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out; mkdir -p $out
cd $R
timeout 300 python -m pytest tests/test_gpu_kernel_forms.py -q -x -k tensor 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pt
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt -- python $R/tools/bench_tensor.py ${1:-256} > $out/r5_tensor.log 2>&1
cat $out/r5_tensor.log | grep "cc="
f=$(find /tmp/pt -name '*kernel_stats.csv' | head -1)
grep -i "tensor" $f | cut -c1-60,150-260 | head -20
cp $f $out/r5_tensor_kernel_stats.csv
