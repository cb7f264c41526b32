#!/bin/bash
# round 2, run AL: same-box A/B: first epilogue column's operands requested before the k-loop (MD_EPI_PREFETCH 1) or not (0)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 1 0 1; do
  MD_EXTRA_FLAGS=-DMD_EPI_PREFETCH=$v bash magicdance_amd/csrc/build.sh > /dev/null 2>&1
  echo "== MD_EPI_PREFETCH=$v"
  timeout 100 python tools/launch_floor.py 2>&1 | grep "md_igemm"
  timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  timeout 300 python bench.py --frames-per-gpu 8 --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
done > gpurun_out/r2al_prefetch_ab.txt 2>&1
MD_EXTRA_FLAGS=-DMD_EPI_PREFETCH=1 bash magicdance_amd/csrc/build.sh > /dev/null 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "igemm" 2>&1 | tail -2 >> gpurun_out/r2al_prefetch_ab.txt
cat gpurun_out/r2al_prefetch_ab.txt
