#!/bin/bash
# round 2, run S: software-pipelined 32x32x16 k-loop: correctness (m32 tests) + timing vs the 16-row tiles
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "m32 or (igemm_conv and (34 or 35 or 36 or 37)) or second_parameter" 2>&1 | tail -5 > gpurun_out/r2s_m32_tests.log
PARTS_CFGS=12,25,34,35,36,37 timeout 300 python tools/igemm_parts.py 2>&1 | grep dbg > gpurun_out/r2s_igemm_parts.txt
tail -3 gpurun_out/r2s_m32_tests.log; cat gpurun_out/r2s_igemm_parts.txt
