#!/bin/bash
# Run ON THE GPU BOX: one step's launches in order with their durations (a compact listing under gpurun_out/onerow_tr/).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/onerow_tr -o t -- python3 $R/tools/prof_step.py --frames ${1:-512} --steps 2 --warmup 1 > $R/gpurun_out/onerow_tr.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob('$R/gpurun_out/onerow_tr/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(rows)//3
out=open('$R/gpurun_out/onerow_tr/last_step.txt','w')
for r in rows[-n:]:
    out.write("%8.1f us grid %8s wg %4s  %s\n"%((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3, r['Grid_Size_X'], r['Workgroup_Size_X'], r['Kernel_Name'][:120]))
PY
rm -f $R/gpurun_out/onerow_tr/*.csv
