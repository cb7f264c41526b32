#!/bin/bash
# round 4, run V: validation of the final tree -- whole GPU test tier, smoke, the default bench line, 768^2 fp16 / fp8 lines
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_*.log
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 --timeout-method=thread 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4v_gpu_tests.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | grep smoke | tee gpurun_out/r4v_smoke.txt
timeout 900 python bench.py 2>gpurun_out/r4v_bench_err.txt | tail -1 > gpurun_out/r4v_bench_n1.json; python -c "
import json; d=json.load(open('gpurun_out/r4v_bench_n1.json')); print('bench', round(d['value'],4), 'frames/s', round(d['ms_per_step'],1), 'ms; igemm frac', round(d['roofline']['frac'],4), 'attention frac', round(d['roofline_attention']['frac'],4), '; configs[2]', round(d['extra']['configs[2]']['value'],3), round(d['extra']['configs[2]']['roofline']['frac'],4))"
for fl in "" "--fp8-attention"; do
  timeout 600 python bench.py --size 96 $fl --no-cpu-baseline --no-extra --steps 3 --warmup 1 2>/dev/null | tail -1 > "gpurun_out/r4v_bench_768${fl:+_fp8}.json"
  python -c "
import json; d=json.load(open('gpurun_out/r4v_bench_768${fl:+_fp8}.json')); print('768x768 [$fl]', round(d['value'],4), 'frames/s', round(d['ms_per_step'],1), 'ms; attention', round(d['roofline_attention']['achieved']), 'TF')"
done
