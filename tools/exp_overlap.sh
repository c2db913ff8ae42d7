#!/bin/bash
# kernel trace of tools/exp_two_streams.py: how much of the two-stream step's kernel time overlaps in time?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ovl -o t -- python3 $R/tools/exp_two_streams.py > /dev/null 2>&1
python3 - <<PY
import csv, glob
rows=[]
for p in glob.glob("$R/gpurun_out/ovl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id") or r.get("Queue_Id")))
rows.sort()
# last 40% of the trace = the two-stream steps
lo = rows[int(len(rows)*0.6)][0]
sel=[r for r in rows if r[0]>=lo]
busy=sum(e-s for s,e,_,_ in sel)
# union length
cur_s,cur_e=sel[0][0],sel[0][1]; union=0
for s,e,_,_ in sel[1:]:
    if s>cur_e: union+=cur_e-cur_s; cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
union+=cur_e-cur_s
print("kernels",len(sel),"sum of durations %.2f ms, union %.2f ms, overlapped %.1f %%"%(busy/1e6,union/1e6,100*(busy-union)/busy))
# which pairs overlap most
import collections
ov=collections.Counter()
active=[]
for s,e,n,q in sel:
    active=[a for a in active if a[1]>s]
    for a in active:
        o=min(e,a[1])-s
        if o>0: ov[(a[2][:45],n[:45])]+=o
    active.append((s,e,n,q))
for k,v in ov.most_common(8): print("%8.2f ms  %s  ||  %s"%(v/1e6,k[0],k[1]))
PY
rm -rf $R/gpurun_out/ovl
