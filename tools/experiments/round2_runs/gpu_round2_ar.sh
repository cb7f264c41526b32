#!/bin/bash
# round 2, run AR: A-stationary tiles: parity test, then the short-K shapes of an 8-frame step timed with them as extra candidates
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "astat" 2>&1 | tail -3 > gpurun_out/r2ar_astat.txt
TUNE_ASTAT=1 TUNE_FILTER=shortk timeout 300 python tools/tune_igemm.py gpurun_out/igemm_tuned_astat.inc 8 2>&1 | grep "^    {" | cut -c1-220 >> gpurun_out/r2ar_astat.txt
cat gpurun_out/r2ar_astat.txt
