#!/bin/bash
# round 2, run O: per-network pass times incl. the merged pass; kernel-trace stats merged vs separate
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R" && mkdir -p gpurun_out/prof_r2o
export TMPDIR=/tmp
timeout 500 python bench.py --no-extra --no-cpu-baseline > gpurun_out/r2o_bench.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2o -o merged --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2o/merged.log 2>&1
MD_MERGE_POSE=0 timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2o -o separate --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2o/separate.log 2>&1
cd $R
rm -f gpurun_out/prof_r2o/*kernel_trace.csv gpurun_out/prof_r2o/*agent_info.csv
ls -la gpurun_out/prof_r2o
grep -o '"unet_ms_per_step": {.*' gpurun_out/r2o_bench.log | cut -c1-900
grep -o '"ddim_step": {[^}]*}' gpurun_out/r2o_bench.log
