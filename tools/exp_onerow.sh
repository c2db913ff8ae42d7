R=$GRAFT_REPO_ROOT
cd $R
for i in 1 2 3; do python3 tools/prof_step.py --frames 512 --steps 20 --warmup 5 2>/dev/null | cut -c1-60; done
python3 tools/prof_step.py --frames 313 --steps 20 --warmup 5 2>/dev/null | cut -c1-60
python3 tools/bench_finetune.py 2>/dev/null | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/onerow -o t -- python3 $R/tools/prof_step.py --frames 512 --steps 6 --warmup 2 > $R/gpurun_out/onerow.log 2>&1
find $R/gpurun_out/onerow -name "*kernel_trace.csv" -delete; find $R/gpurun_out/onerow -name "*agent_info.csv" -delete
