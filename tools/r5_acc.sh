#!/bin/bash
# round 5: `sol += cor` inside the last sweep of a V-cycle (ACC) -- tests, then the three workloads with it on and off
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out; mkdir -p $out
cd $R
timeout 1500 python -m pytest tests/test_gpu_cf_abec.py tests/test_gpu_sensitivity.py tests/test_gpu_rb_nbr.py tests/test_gpu_kernel_forms.py tests/test_gpu_ns.py tests/test_gpu_tensor_bottom.py tests/test_gpu_walls_inkernel.py -x -q > $out/r5acc_tests.txt 2>&1
tail -4 $out/r5acc_tests.txt
for m in 0 1; do
IAMRX_MG_ACC_LAST_SWEEP=$m timeout 600 python tools/run_steps.py > $out/r5acc_tg_$m.txt 2>&1
echo "TG acc=$m $(grep 'ms/step' $out/r5acc_tg_$m.txt)"
IAMRX_MG_ACC_LAST_SWEEP=$m timeout 600 python tools/run_ldc_steps.py > $out/r5acc_ldc_$m.txt 2>&1
echo "LDC acc=$m $(grep 'ms/step' $out/r5acc_ldc_$m.txt)"
IAMRX_MG_ACC_LAST_SWEEP=$m timeout 600 python tools/bench_amr.py 256 3 > $out/r5acc_amr_$m.json 2> $out/r5acc_amr_$m.err
python - <<P
import json
d = json.loads(open("gpurun_out/r5acc_amr_$m.json").read().strip().splitlines()[-1])
print("AMR acc=$m", d["ms_per_coarse_step"], d["cells_advanced_per_sec"])
P
done
