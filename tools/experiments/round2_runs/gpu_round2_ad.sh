#!/bin/bash
# round 2, run AD: kernel-trace stats after the epilogue changes (compare with profiles/round2_kernel_stats.txt) + repeat bench
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R" && mkdir -p gpurun_out/prof_r2ad
export TMPDIR=/tmp
timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2ad_bench.log 2>&1
timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2ad_bench_b.log 2>&1
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2ad -o kt --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2ad/kt.log 2>&1
cd $R
rm -f gpurun_out/prof_r2ad/*_kernel_trace.csv
python tools/summarize_profiles.py gpurun_out/prof_r2ad gpurun_out/r2ad_summary
for f in r2ad_bench r2ad_bench_b; do grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1; done
head -30 gpurun_out/r2ad_summary_kernel_stats.txt | cut -c1-120
