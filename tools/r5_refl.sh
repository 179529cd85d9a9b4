#!/bin/bash
# round 5: mirror images at Neumann walls inside the nodal wrap kernels -- tests, then LidDrivenCavity 256^3 with the switch on and off
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out; mkdir -p $out
cd $R
timeout 1500 python -m pytest tests/test_gpu_nodal_gsr.py -x -q -k "mirror_images" > $out/r5refl_tests.txt 2>&1
tail -4 $out/r5refl_tests.txt
timeout 1500 python -m pytest tests/test_gpu_walls.py tests/test_gpu_ldc.py tests/test_gpu_nodal_dirichlet.py tests/test_gpu_inflow_outflow.py tests/test_gpu_hydrostatic.py tests/test_gpu_slab_mg.py -x -q > $out/r5refl_tests2.txt 2>&1
tail -3 $out/r5refl_tests2.txt
for m in 0 1 0 1; do
IAMRX_NODAL_REFLECT_WRAP=$m timeout 600 python tools/run_ldc_steps.py > $out/r5refl_ldc_$m.txt 2>&1
echo "LDC reflect=$m $(grep 'ms/step' $out/r5refl_ldc_$m.txt)"
done
