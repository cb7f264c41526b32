#!/bin/bash
# round 4, run A: the ring form of md_igemm -- parity first, then the tuner against the committed 2-stage table (M <= 1024)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_igemm_ring.py -m gpu -q --maxfail=40 --timeout 120 --timeout-method=thread 2>&1 | tail -25 > gpurun_out/r4a_ring_tests.txt
cat gpurun_out/r4a_ring_tests.txt | tail -8
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k igemm --timeout 120 --timeout-method=thread 2>&1 | tail -5 > gpurun_out/r4a_legacy_tests.txt
cat gpurun_out/r4a_legacy_tests.txt
timeout 600 python tools/tune_ring.py gpurun_out/r4a_tuned.inc --mmax 1024 > gpurun_out/r4a_tune_ring.txt 2>&1
tail -70 gpurun_out/r4a_tune_ring.txt
