#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_gpu_godunov_fused.py tests/test_gpu_godunov.py tests/test_gpu_walls.py tests/test_gpu_ldc.py tests/test_gpu_kernel_forms.py tests/test_gpu_diffusion_ops.py -q -x 2>&1 | tail -5
bash tools/r5_state.sh
