#!/bin/bash
# GPU box: round-2 run F -- full GPU test tier, bench A/B of the fused pose residuals, step breakdown, rocprofv3 kernel stats + PMC traffic
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/prof_r2
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r2f_pytest.log
timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extra > gpurun_out/r2f_bench.log 2>&1
MD_FUSE_POSE=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extra > gpurun_out/r2f_bench_nofuse.log 2>&1
timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r2f_step_breakdown_1frame.txt 2>&1
timeout 300 python tools/step_breakdown.py 8 > gpurun_out/r2f_step_breakdown_8frames.txt 2>&1
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2 -o kt --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2/kt.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r2 -o pmc_fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r2 -o pmc_write --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2/pmc_write.log 2>&1
cd $R
rm -f gpurun_out/prof_r2/*_kernel_trace.csv   # per-dispatch rows: large; the stats + counter csv are what gets summarised
ls -la gpurun_out/prof_r2 | head -20
tail -4 gpurun_out/r2f_pytest.log
for f in r2f_bench r2f_bench_nofuse; do grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1; done
head -3 gpurun_out/r2f_step_breakdown_1frame.txt | tail -2
