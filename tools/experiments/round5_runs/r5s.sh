#!/bin/bash
# round 5, run S: failing cases of run R in detail
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_igemm_gn.py -q --timeout 600 --tb=line 2>&1 | grep -E "differs|declined|Error|passed|failed" | cut -c1-420 | tee gpurun_out/r5s_tests.txt | tail -45
