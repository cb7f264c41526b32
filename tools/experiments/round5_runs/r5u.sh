#!/bin/bash
# round 5, run U: parity suites + the new repeatability tests on the library without packed fp32 in the GEMM units
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_fullsize.log gpurun_out/parity_e2e.log gpurun_out/parity_vae.log
timeout 1500 python -m pytest tests/test_gpu_repeatability.py tests/test_gpu_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_vae.py -q --timeout 900 --tb=short 2>&1 | tail -30 | cut -c1-300 | tee gpurun_out/r5u_tests.txt
grep -E "final latent|worst ratio|ratio" gpurun_out/parity_fullsize.log | tail -12 | cut -c1-300
