#!/bin/bash
# round 4, run F: the 8-wave ring tiles (configs 53-57): parity, then the tuner on the large-M layers
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_igemm_ring.py -m gpu -q --maxfail=20 --timeout 120 --timeout-method=thread -k "53 or 54 or 55 or 56 or 57 or table" 2>&1 | tail -12 > gpurun_out/r4f_ring_tests.txt
tail -6 gpurun_out/r4f_ring_tests.txt
timeout 900 python tools/tune_ring.py gpurun_out/r4f_tuned.inc --mmin 4096 --mmax 300000 --cfgs 47,48,49,53,54,55,56,57 > gpurun_out/r4f_tune_ring.txt 2>&1
tail -60 gpurun_out/r4f_tune_ring.txt
