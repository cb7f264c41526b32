#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -k "igemm or merged_and_forked or (golden and small_b2)" 2>&1 | tail -12 > gpurun_out/r2aq_tests.log
cat gpurun_out/r2aq_tests.log
