#!/bin/bash
# round 2, run U: refresh the judged profiles for the final launch mix (merged pass): kernel-trace stats, PMC fetch/write, step breakdown
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R" && mkdir -p gpurun_out/prof_r2u
export TMPDIR=/tmp
timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r2u_step_breakdown_1frame.txt 2>&1
timeout 300 python tools/step_breakdown.py 8 > gpurun_out/r2u_step_breakdown_8frames.txt 2>&1
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2u -o kt --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2u/kt.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r2u -o pmc_fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2u/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r2u -o pmc_write --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2u/pmc_write.log 2>&1
cd $R
rm -f gpurun_out/prof_r2u/*_kernel_trace.csv
python tools/summarize_profiles.py gpurun_out/prof_r2u gpurun_out/r2u_summary
ls gpurun_out/prof_r2u gpurun_out | head -40; head -3 gpurun_out/r2u_step_breakdown_1frame.txt; head -3 gpurun_out/r2u_step_breakdown_8frames.txt
