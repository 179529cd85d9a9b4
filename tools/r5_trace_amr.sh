#!/bin/bash
# kernel trace of the AMR 256^3 + 256^3 workload, reduced to the LAST coarse step as a compact CSV (gpurun_out/r5_amr_trace.csv)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/ptr
AMR_N0=${AMR_N0:-256} timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ptr -- python $R/tools/${DBG:-run_amr_steps.py} > /tmp/ptr.log 2>&1
grep 'ms/step' /tmp/ptr.log
f=$(find /tmp/ptr -name '*kernel_trace.csv' | head -1)
python3 - "$f" "$R/gpurun_out/${OUT:-r5_amr_trace.csv}" <<'PY'
import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
def tiny(r): return 'k_fill' in r['Kernel_Name'] and int(r['Grid_Size_X'])<=512 and int(r['Grid_Size_Y'])==1
idx=[i for i in range(len(rows)-2) if tiny(rows[i]) and tiny(rows[i+1]) and tiny(rows[i+2])]
a,b=idx[-2]+2,idx[-1]
seg=rows[a+1:b]
n=len(seg)//4
seg=seg[3*n:]
w=csv.writer(open(sys.argv[2],'w'))
w.writerow(['Kernel_Name','Grid_Size_X','Grid_Size_Y','Grid_Size_Z','Start_Timestamp','End_Timestamp'])
for r in seg:
    nm=re.sub(r'\(.*','',r['Kernel_Name']).replace('void ','')[:110]
    w.writerow([nm,r['Grid_Size_X'],r['Grid_Size_Y'],r['Grid_Size_Z'],r['Start_Timestamp'],r['End_Timestamp']])
print('rows',len(seg))
PY
