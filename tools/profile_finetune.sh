#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/profile_finetune.sh <tag>
# Kernel stats of the batch-of-one fine-tune step (tools/bench_finetune.py, T = 1876): kernel time against wall time.
set -u
TAG=${1:-ft}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 python3 $R/tools/bench_finetune.py > $OUT/finetune.json 2> $OUT/finetune.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $R/tools/bench_finetune.py > $OUT/stats.log 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/finetune.json
