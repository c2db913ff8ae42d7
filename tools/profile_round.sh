#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash tools/profile_round.sh <tag>
# Produces under gpurun_out/prof_<tag>/: kernel stats of `bench.py`, the default bench line, and four separate
# --pmc passes (counters are never combined with other trace domains), then the per-kernel PMC summary.
set -u
TAG=${1:-final}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python3 $R/bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras > $OUT/stats.log 2>&1
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc$i -o pmc -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/pmc$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/pmc_summary.csv
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# keep the merge-back small: drop the raw traces
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
ls -la $OUT
