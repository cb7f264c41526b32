#!/bin/bash
# round 4, run S: validation of the tree -- whole GPU test tier, smoke, the default bench line, step breakdowns (1 / 8 frames),
# rocprofv3 kernel stats + PMC traffic passes (tools/run_profiles.sh)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_*.log
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 --timeout-method=thread 2>&1 | tail -5 > gpurun_out/r4s_gpu_tests.txt; cat gpurun_out/r4s_gpu_tests.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | grep smoke | tee gpurun_out/r4s_smoke.txt
timeout 900 python bench.py 2>gpurun_out/r4s_bench_err.txt | tail -1 > gpurun_out/r4s_bench_n1.json; python -c "
import json; d=json.load(open('gpurun_out/r4s_bench_n1.json')); print('bench', round(d['value'],4), 'frames/s', round(d['ms_per_step'],1), 'ms; igemm frac', round(d['roofline']['frac'],4), 'traffic', d['roofline']['traffic'], '; configs[2]', d['extra']['configs[2]']['value'], d['extra']['configs[2]'].get('roofline',{}).get('frac'))"
timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r4s_step_breakdown_1frame.txt 2>&1; head -3 gpurun_out/r4s_step_breakdown_1frame.txt | tail -2
timeout 300 python tools/step_breakdown.py 8 > gpurun_out/r4s_step_breakdown_8frames.txt 2>&1; head -3 gpurun_out/r4s_step_breakdown_8frames.txt | tail -2
bash tools/run_profiles.sh r4s_prof 2>&1 | tail -8
python tools/summarize_profiles.py gpurun_out/r4s_prof gpurun_out/r4s_summary 2>&1 | tail -3
