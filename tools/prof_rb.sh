#!/bin/bash
# per-kernel times of tools/bench_rb.py under rocprofv3 (scratch tool)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prb && rocprofv3 --kernel-trace --stats -d /tmp/prb -o rb --output-format csv -- python $R/tools/bench_rb.py "$@" 2>&1 | grep -E "^n="
python - <<'PY'
import csv, glob
f = glob.glob('/tmp/prb/**/rb_kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'gsrb' in r['Name']: print(f"{float(r['AverageNs'])/1e3:9.1f} us x {r['Calls']:>5}  {r['Name'][:90]}")
PY
