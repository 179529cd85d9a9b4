#!/bin/bash
# host side of the uniform 256^3 step: HIP runtime API statistics of tools/run_steps.py (rocprofv3 --hip-runtime-trace --stats)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ph
timeout 600 rocprofv3 --hip-runtime-trace --stats --output-format csv -d /tmp/ph -- python $R/tools/run_steps.py > /tmp/ph.log 2>&1
grep "ms/step" /tmp/ph.log
f=$(find /tmp/ph -name '*hip_api_stats.csv' | head -1)
head -14 $f | cut -c1-150
