#!/bin/bash
# round 5: the sweep kernel with in-kernel coarse/fine faces -- parity tests, then the AMR 256^3 + 256^3 workload with it on and off
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out; mkdir -p $out
cd $R
timeout 900 python -m pytest tests/test_gpu_kernel_forms.py -x -q -k "refined_box" > $out/r5cf_tests.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_cf_abec.py tests/test_gpu_cf_tensor.py tests/test_gpu_amr_step.py -x -q >> $out/r5cf_tests.txt 2>&1
tail -5 $out/r5cf_tests.txt
IAMRX_GSRB_RB_CF=0 timeout 600 python tools/bench_amr.py 256 3 > $out/r5cf_amr_off.json 2> $out/r5cf_amr_off.err
IAMRX_GSRB_RB_CF=1 timeout 600 python tools/bench_amr.py 256 3 > $out/r5cf_amr_on.json 2> $out/r5cf_amr_on.err
python - <<'P'
import json
for t in ("off", "on"):
    try:
        d = json.loads(open(f"gpurun_out/r5cf_amr_{t}.json").read().strip().splitlines()[-1])
        print(t, d["ms_per_coarse_step"], d["cells_advanced_per_sec"], d.get("sections_ms_per_coarse_step", {}).get("advance_level1"), d.get("sync_project_iters"), d.get("mac_sync_iters"))
    except Exception as e:
        print(t, "failed", e)
P
NAMELEN=90 NTOP=30 AMR_N0=256 DBG=run_amr_steps.py bash tools/profile_step.sh > $out/r5cf_step_amr256.txt 2>&1
head -34 $out/r5cf_step_amr256.txt
