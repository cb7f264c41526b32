#!/bin/bash
# round 5, run G: md_ff_block step-schedule variants (fatter steps for the 32- / 64-row workgroups): parity + microbenchmark
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ffblock.py -q -x --timeout 300 2>&1 | tail -5 | tee gpurun_out/r5g_ffblock_tests.txt
timeout 300 python tools/ffblock_bench.py 320,2,4096 320,3,4096 320,4,4096 320,6,4096 320,16,4096 > gpurun_out/r5g_ffblock_bench.txt 2>&1; tail -12 gpurun_out/r5g_ffblock_bench.txt
