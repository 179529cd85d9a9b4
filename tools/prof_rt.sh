#!/bin/bash
# rocprofv3 kernel statistics of the C5 workload (tools/run_rt.py): top kernels by total time + total kernel time against the wall time
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
out=$R/gpurun_out; mkdir -p $out
rm -rf /tmp/prt
RT_STEPS=${RT_STEPS:-1} timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prt -- python $R/tools/run_rt.py > $out/${TAG:-r6_rt}_prof.log 2>&1
f=$(find /tmp/prt -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY' > $out/${TAG:-r6_rt}_kernel_stats.txt
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel time s", tot/1e9, "kernels", sum(int(r['Calls']) for r in rows))
for r in sorted(rows,key=lambda r:-float(r['TotalDurationNs']))[:45]:
    nm=re.sub(r'\(.*','',r['Name']).replace('void ','')[:90]
    print("%9.1f ms %8d x %9.1f us  %s"%(float(r['TotalDurationNs'])/1e6,int(r['Calls']),float(r['AverageNs'])/1e3,nm))
PY
head -50 $out/${TAG:-r6_rt}_kernel_stats.txt
tail -3 $out/${TAG:-r6_rt}_prof.log
