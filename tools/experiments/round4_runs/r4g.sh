#!/bin/bash
# round 4, run G: register-pipelined ring loop (configs 58-64): parity, then against the plain ring and the 2-stage table (M >= 4096)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_igemm_ring.py -m gpu -q --maxfail=20 --timeout 120 --timeout-method=thread -k "58 or 59 or 60 or 61 or 62 or 63 or 64" 2>&1 | tail -12 > gpurun_out/r4g_ring_tests.txt
tail -6 gpurun_out/r4g_ring_tests.txt
timeout 900 python tools/tune_ring.py gpurun_out/r4g_tuned.inc --mmin 4096 --mmax 300000 --cfgs 49,53,54,55,56,57,58,59,60,61,62,63,64 > gpurun_out/r4g_tune_ring.txt 2>&1
grep -v "^/opt" gpurun_out/r4g_tune_ring.txt | cut -c1-260
