#!/bin/bash
# round 5, run V: validation of the FINAL tree (GEMM units without packed fp32, GroupNorm in the split-K reduction) (whole GPU test tier, smoke), the default bench line, kernel-trace statistics, per-launch
# step breakdowns (1 / 8 frames)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out/prof_r5v
export TMPDIR=/tmp
R=$(pwd)
rm -f gpurun_out/parity_*.log
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 --timeout-method=thread 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" | tail -12 | tee gpurun_out/r5v_gpu_tests.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | grep smoke | tee gpurun_out/r5v_smoke.txt
timeout 900 python bench.py 2>gpurun_out/r5v_bench_err.txt | tail -1 > gpurun_out/r5v_bench_n1.json; python -c "
import json; d=json.load(open('gpurun_out/r5v_bench_n1.json')); print('bench', round(d['value'],4), 'frames/s', round(d['ms_per_step'],1), 'ms; igemm frac', round(d['roofline']['frac'],4), 'attention frac', round(d['roofline_attention']['frac'],4), '; configs[2]', round(d['extra']['configs[2]']['value'],3), round(d['extra']['configs[2]']['roofline']['frac'],4), 'cpu', d['cpu_baseline']['cores'], d['cpu_baseline'].get('threads_probe_s'))" | tee gpurun_out/r5v_bench.txt
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r5v -o kt --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline > $R/gpurun_out/prof_r5v/bench_kt.log 2>&1; echo kt rc=$?)
rm -f gpurun_out/prof_r5v/kt_kernel_trace.csv
python tools/summarize_profiles.py gpurun_out/prof_r5v gpurun_out/r5v > /dev/null 2>&1; head -12 gpurun_out/r5v_kernel_stats.txt | cut -c1-160
timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r5v_step_breakdown_1frame.txt 2>&1; head -3 gpurun_out/r5v_step_breakdown_1frame.txt | tail -2
timeout 300 python tools/step_breakdown.py 8 > gpurun_out/r5v_step_breakdown_8frames.txt 2>&1; head -3 gpurun_out/r5v_step_breakdown_8frames.txt | tail -2
timeout 300 python tools/repeat_check.py 6 1 2>&1 | tail -2 | tee gpurun_out/r5v_repeat_check.txt
