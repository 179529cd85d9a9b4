#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
nproc
for t in 4 16 default; do
  if [ "$t" = default ]; then unset OMP_NUM_THREADS; else export OMP_NUM_THREADS=$t; fi
  echo "OMP_NUM_THREADS=$t"
  timeout 600 python -m pytest "tests/test_gpu_c3.py::test_double_shear_layer_slab_matches_the_oracle" "tests/test_gpu_ns.py::test_advance_at_config_c1_size_matches_oracle" -q -x --durations=5 2>&1 | grep "s call\|passed\|failed"
done
