#!/bin/bash
# Per-step kernel time by kernel and launch grid: rocprofv3 kernel trace of tools/run_steps.py, reduced to the four steps between its
# two marker launches.  Run on the GPU box from the repo root:  bash tools/profile_step.sh   (DBG=<other driver script in tools/> to change the workload)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/pstep
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pstep -- python $R/tools/${DBG:-run_steps.py} > /tmp/pstep.log 2>&1
grep 'ms/step' /tmp/pstep.log
f=$(find /tmp/pstep -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv,re,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# markers: k_fill with grid for 343 cells -> find the two
# markers: three consecutive tiny k_fill launches (run_steps.py / run_amr_steps.py issue them in front of and behind the timed steps)
def tiny(r): return 'k_fill' in r['Kernel_Name'] and int(r['Grid_Size_X'])<=512 and int(r['Grid_Size_Y'])==1
idx=[i for i in range(len(rows)-2) if tiny(rows[i]) and tiny(rows[i+1]) and tiny(rows[i+2])]
a,b=idx[-2]+2,idx[-1]
seg=rows[a+1:b]
tot=collections.Counter(); cnt=collections.Counter()
for r in seg:
    nm=re.sub(r'\(.*','',r['Kernel_Name']).replace('void ','')[:int(__import__('os').environ.get('NAMELEN','60'))]+' g='+r['Grid_Size_X']+'x'+r['Grid_Size_Y']+'x'+r['Grid_Size_Z']
    d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    tot[nm]+=d; cnt[nm]+=1
T=sum(tot.values())
span=int(seg[-1]['End_Timestamp'])-int(seg[0]['Start_Timestamp'])
print('kernel ms/step', T/4e6, 'span ms/step', span/4e6, 'launches/step', len(seg)/4)
for nm,v in tot.most_common(int(__import__('os').environ.get('NTOP','60'))):
    print("%7.3f ms %6.1f x %8.1f us  %s"%(v/4e6, cnt[nm]/4, v/cnt[nm]/1e3, nm))
PY
