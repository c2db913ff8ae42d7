#!/bin/bash
# Run ON THE GPU BOX from the repo root:  bash tools/profile_shape.sh <tag> <frames> [batch]
# Kernel stats + MFMA-busy counters of the fused pretext step at another crop length (tools/prof_step.py).
set -u
TAG=${1:-T313}; FR=${2:-313}; BA=${3:-16}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 python3 $R/tools/prof_step.py --frames $FR --batch $BA --steps 20 --warmup 3 > $OUT/step.json 2> $OUT/step.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $R/tools/prof_step.py --frames $FR --batch $BA > $OUT/stats.log 2>&1
i=0
for SET in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc$i -o pmc -- python3 $R/tools/prof_step.py --frames $FR --batch $BA --steps 2 --warmup 1 > $OUT/pmc$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/pmc_summary.csv
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
cat $OUT/step.json
