#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for m in -1 0; do
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ftw_$m -o t -- python3 $R/tools/bench_finetune.py --win128 $m > /dev/null 2>&1
  python3 $R/tools/prof_summary.py $R/gpurun_out/ftw_$m 170 > $R/gpurun_out/ftw_$m.txt
  rm -rf $R/gpurun_out/ftw_$m
done
