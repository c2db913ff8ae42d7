#!/bin/bash
# Run ON THE GPU BOX: bash tools/exp_trace.sh <tag> [pattern]  -- per-dispatch durations of one bench step, filtered
TAG=${1:-x}; PAT=${2:-wgrad_core}
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o t -- python3 $R/bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-extras > $R/gpurun_out/prof_$TAG.log 2>&1
python3 $R/tools/trace_step.py $R/gpurun_out/prof_$TAG > $R/gpurun_out/trace_$TAG.txt
find $R/gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
grep -E "$PAT" $R/gpurun_out/trace_$TAG.txt | cut -c1-120
tail -1 $R/gpurun_out/trace_$TAG.txt
grep -o '"ms_per_step": [0-9.]*' $R/gpurun_out/prof_$TAG.log
