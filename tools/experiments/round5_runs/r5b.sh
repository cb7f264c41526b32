#!/bin/bash
# round 5, run B: md_ff_block with the per-workgroup chunk rotation (L2 channel hot-spotting fix): parity + microbenchmark
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ffblock.py -q -x --timeout 300 2>&1 | tail -5 | tee gpurun_out/r5b_ffblock_tests.txt
timeout 300 python tools/ffblock_bench.py > gpurun_out/r5b_ffblock_bench.txt 2>&1; tail -12 gpurun_out/r5b_ffblock_bench.txt
