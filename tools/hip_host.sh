#!/bin/bash
# host side of a step: HIP runtime API statistics (rocprofv3 --hip-runtime-trace --stats) of tools/${DBG:-run_steps.py} and the host's
# launch-to-launch cadence between synchronisation points (round 5: tools/r5_hip.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ph
timeout 600 rocprofv3 --hip-runtime-trace --stats --output-format csv -d /tmp/ph -- python $R/tools/${DBG:-run_steps.py} > /tmp/ph.log 2>&1
grep "ms/step" /tmp/ph.log
f=$(find /tmp/ph -name '*hip_api_stats.csv' | head -1)
head -6 $f | cut -c1-110
t=$(find /tmp/ph -name '*hip_api_trace.csv' | head -1)
python3 - "$t" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
gaps=[]; prev=None
for r in rows:
    f=r['Function']
    if f=='hipLaunchKernel':
        if prev is not None: gaps.append(int(r['Start_Timestamp'])-prev)
        prev=int(r['Start_Timestamp'])
    elif 'Synchronize' in f or f.startswith('hipMemcpy'):
        prev=None
gaps.sort()
n=len(gaps)
print('launch-to-launch host cadence (no sync in between): n=%d median %.1f us, p25 %.1f, p75 %.1f, p90 %.1f, mean %.1f us'%(n,gaps[n//2]/1e3,gaps[n//4]/1e3,gaps[3*n//4]/1e3,gaps[9*n//10]/1e3,sum(gaps)/n/1e3))
PY
