#!/bin/bash
# round 5, run N: validation of the FINAL tree: whole GPU test tier, smoke, the default bench line (+ 768x768 fp16 / fp8 lines)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_*.log
timeout 2400 python -m pytest tests -m gpu -q --timeout 900 --timeout-method=thread 2>&1 | grep -E "passed|failed|error|FAILED|ERROR" | tail -12 | tee gpurun_out/r5n_gpu_tests.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | grep smoke | tee gpurun_out/r5n_smoke.txt
timeout 900 python bench.py 2>gpurun_out/r5n_bench_err.txt | tail -1 > gpurun_out/r5n_bench_n1.json; python -c "
import json; d=json.load(open('gpurun_out/r5n_bench_n1.json')); print('bench', round(d['value'],4), 'frames/s', round(d['ms_per_step'],1), 'ms; igemm frac', round(d['roofline']['frac'],4), 'traffic', d['roofline']['traffic'], d['roofline']['traffic_unit'][:60], 'attention frac', round(d['roofline_attention']['frac'],4), '; configs[2]', round(d['extra']['configs[2]']['value'],3), round(d['extra']['configs[2]']['roofline']['frac'],4), d['extra']['configs[2]']['roofline']['traffic'], 'cpu', d['cpu_baseline']['cores'], d['cpu_baseline']['value'])" | tee gpurun_out/r5n_bench.txt
for fp8 in "" "--fp8-attention"; do
  timeout 600 python bench.py --size 96 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra $fp8 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('768x768 $fp8', round(d['value'],4), 'frames/s', round(d['ms_per_step'],1), 'ms')"
done 2>&1 | tee gpurun_out/r5n_768.txt
