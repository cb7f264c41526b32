#!/bin/bash
# round 3, run G: same-box A/B of the attention v3 instruction order (fragment reads ahead of the LDS-DMA issue block, early V reads)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
{
for v in 0 1; do
  MD_EXTRA_FLAGS=-DMD_ATTN_R3=$v bash magicdance_amd/csrc/build.sh > /dev/null 2>&1
  echo "== MD_ATTN_R3=$v"
  ATTN_BENCH_ONLY=0,1,2,7,8 timeout 200 python tools/attn_bench.py 2>&1 | grep TF
  ATTN_BENCH_ONLY=0,1,2,7,8 timeout 200 python tools/attn_bench.py 2>&1 | grep TF
  timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  timeout 300 python bench.py --frames-per-gpu 8 --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
done
echo "== tests (R3=1 build)"; timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "attention" 2>&1 | tail -2
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x 2>&1 | tail -2
} > gpurun_out/r3g.txt 2>&1
cat gpurun_out/r3g.txt
