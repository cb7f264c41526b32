#!/bin/bash
# round 5, run C: ablation of md_ff_block (which part costs the ~1 us per ring step): no s1/s0 DMA | no DMA | no MFMA | neither
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for a in 1 2 4 6; do
  echo "== ablate $a"; MD_HIP_LIB=$PWD/tools/bin/libmd_ablate$a.so timeout 200 python tools/ffblock_bench.py 320,2,4096 320,16,4096 2>&1 | grep "C="
done | tee gpurun_out/r5c_ablate.txt
