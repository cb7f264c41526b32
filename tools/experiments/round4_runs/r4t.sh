#!/bin/bash
# round 4, run T: fp8 attention on the pipelined loop (piecewise-linear exp2) -- unit / e2e / 768^2 parity tests of the fp8 path, then
# accuracy + timing against the 2-stage fp8 kernel and the fp16 kernel; 768^2 bench with and without fp8 attention
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -q -x -k "fp8" 2>&1 | grep -E "passed|failed|Error|assert" | tail -6
grep -i fp8 gpurun_out/parity_e2e.log gpurun_out/parity_fullsize.log | tail -6
timeout 600 python tools/attn_fp8_ab.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4t_attn_fp8_ab.txt
for fl in "" "--fp8-attention"; do
  timeout 600 python bench.py --size 96 $fl --no-cpu-baseline --no-roofline --no-extra --steps 3 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('768x768 [$fl]', round(d['value'],4), 'frames/s', round(d['ms_per_step'],1), 'ms')" | tee -a gpurun_out/r4t_attn_fp8_ab.txt
done
