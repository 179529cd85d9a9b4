#!/bin/bash
# round 5: the direct bottom solve of the tensor operator -- tests, then LidDrivenCavity 256^3 and the AMR workload with it on and off
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out; mkdir -p $out
cd $R
timeout 900 python -m pytest tests/test_gpu_tensor_bottom.py -x -q > $out/r5tb_tests.txt 2>&1
tail -15 $out/r5tb_tests.txt
timeout 900 python -m pytest tests/test_gpu_walls.py tests/test_gpu_cf_tensor.py tests/test_gpu_diffusion_ops.py tests/test_gpu_ldc.py tests/test_gpu_walls_inkernel.py -x -q > $out/r5tb_tests2.txt 2>&1
tail -3 $out/r5tb_tests2.txt
for m in 0 1; do
IAMRX_TENSOR_BOTTOM_DIRECT=$m timeout 600 python tools/run_ldc_steps.py > $out/r5tb_ldc_$m.txt 2>&1
grep "ms/step" $out/r5tb_ldc_$m.txt
IAMRX_TENSOR_BOTTOM_DIRECT=$m timeout 600 python tools/bench_amr.py 256 3 > $out/r5tb_amr_$m.json 2> $out/r5tb_amr_$m.err
python - <<P
import json
d = json.loads(open("gpurun_out/r5tb_amr_$m.json").read().strip().splitlines()[-1])
s = d["sections_ms_per_coarse_step"]
print($m, d["ms_per_coarse_step"], d["cells_advanced_per_sec"], {k: (round(v, 1) if not isinstance(v, dict) else "") for k, v in s.items()})
P
done
