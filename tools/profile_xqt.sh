#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash tools/profile_xqt.sh <tag>
# generate_XQT (BASELINE configs[1]): kernel stats of tools/bench_xqt.py (32 / 512 / 4096 clips: the fused kernel, then the
# per-stage launches timed alone) and separate --pmc passes (never combined with other trace domains) at 4096 clips.
set -u
TAG=${1:-xqt}
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 python3 $R/tools/bench_xqt.py --cpu > $OUT/bench_xqt.json 2> $OUT/bench_xqt.err
timeout 300 python3 $R/tools/bench_xqt_sizes.py > $OUT/bench_xqt_sizes.json 2>> $OUT/bench_xqt.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $R/tools/bench_xqt.py > $OUT/stats.log 2>&1
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $SET --output-format csv -d $OUT/pmc$i -o pmc -- python3 $R/tools/bench_xqt.py --only 4096 > $OUT/pmc$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $OUT > $OUT/pmc_summary.csv
python3 $R/tools/prof_summary.py $OUT/stats 90 > $OUT/kernel_summary.txt
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
ls -la $OUT
