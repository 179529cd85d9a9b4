#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 1400 python -m pytest tests/test_gpu_sensitivity.py tests/test_gpu_slab_mg.py tests/test_gpu_syncreg.py tests/test_gpu_temp.py tests/test_gpu_validation.py tests/test_gpu_walls.py tests/test_gpu_walls_inkernel.py tests/test_gpu_ldc.py tests/test_gpu_abec.py tests/test_gpu_poison.py -x -q --durations=25 > gpurun_out/rest.log 2>&1; tail -40 gpurun_out/rest.log
