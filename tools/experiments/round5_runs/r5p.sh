#!/bin/bash
# round 5, run P: the single-pass short-segment attention kernel: parity, microbenchmark A/B (MD_ATTN_SHORT=0 | default), bench A/B
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x --timeout 600 -k "attention" 2>&1 | tail -4 | tee gpurun_out/r5p_tests.txt
{ echo "== MD_ATTN_SHORT=0 (tile-loop kernels)"; MD_ATTN_SHORT=0 timeout 200 python tools/attn_short_bench.py 2>&1 | grep "B="; echo "== attn_short_kernel"; timeout 200 python tools/attn_short_bench.py 2>&1 | grep "B="; } | tee gpurun_out/r5p_attn_short_bench.txt
for i in 1 2; do for v in 0 1; do
  MD_ATTN_SHORT=$v timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d.get('extra',{}).get('configs[2]',{}); print('MD_ATTN_SHORT=$v', 'frames/s', round(d['value'],4), 'ms', round(d['ms_per_step'],1), 'configs[2]', round(e.get('value',0),3))"
done; done 2>&1 | tee gpurun_out/r5p_bench_ab.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q --timeout 900 2>&1 | tail -4 | tee -a gpurun_out/r5p_tests.txt
