#!/bin/bash
# A/B of the cv3 / cv4 window length on the fused step, kernel traces of both: tools/exp_win128.sh <frames>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-512}
for m in -1 1; do
  python3 $R/tools/prof_step.py --frames $T --steps 20 --warmup 3 --win128 $m
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/w128_${T}_$m -o t -- python3 $R/tools/prof_step.py --frames $T --steps 6 --warmup 2 --win128 $m > /dev/null 2>&1
  python3 $R/tools/prof_summary.py $R/gpurun_out/w128_${T}_$m 170 > $R/gpurun_out/w128_${T}_$m.txt
  rm -rf $R/gpurun_out/w128_${T}_$m
done
