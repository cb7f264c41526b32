#!/bin/bash
# round 3, run M: tiled weight storage in the engine: kernel tests, e2e parity, same-box A/B
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== kernels"; timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "tiled or igemm_conv" 2>&1 | tail -3
  echo "== e2e"; timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_vae.py -m gpu -q -x 2>&1 | tail -3
  echo "== fullsize"; timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "baseline or batch8" 2>&1 | tail -3
  for env in "MD_W_TILED=1" "MD_W_TILED=0" "MD_W_TILED=1" "MD_W_TILED=0"; do
    echo "== $env"
    env $env timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
    env $env timeout 300 python bench.py --frames-per-gpu 8 --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  done
} > gpurun_out/r3m.txt 2>&1
cat gpurun_out/r3m.txt
