#!/bin/bash
# round 5, run D: md_ff_block with the static piece schedule: parity, microbenchmark, ablation (no DMA | no DMA + no MFMA)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ffblock.py -q -x --timeout 300 2>&1 | tail -5 | tee gpurun_out/r5d_ffblock_tests.txt
timeout 300 python tools/ffblock_bench.py > gpurun_out/r5d_ffblock_bench.txt 2>&1; tail -12 gpurun_out/r5d_ffblock_bench.txt
for a in 2 6; do
  echo "== ablate $a"; MD_HIP_LIB=$PWD/tools/bin/libmd_ablate$a.so timeout 200 python tools/ffblock_bench.py 320,2,4096 320,16,4096 2>&1 | grep "C="
done | tee gpurun_out/r5d_ablate.txt
