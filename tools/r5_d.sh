#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 600 python -m pytest tests/test_gpu_walls_inkernel.py tests/test_gpu_walls.py tests/test_gpu_ldc.py tests/test_gpu_poison.py -q -x 2>&1 | tail -3
export NAMELEN=90 NTOP=12
bash tools/profile_step.sh 2>&1 | head -14
DBG=run_ldc_steps.py bash tools/profile_step.sh 2>&1 | head -12
