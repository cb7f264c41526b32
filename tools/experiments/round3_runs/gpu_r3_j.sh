#!/bin/bash
# round 3, run J: LDS-staged igemm epilogue (16-byte row-major stores): kernel tests, isolated A/B on the epilogue-bound shapes, bench A/B
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== kernels"; timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_vae.py -m gpu -q -x -k "igemm or conv or linear or vae" 2>&1 | tail -3
  for env in "MD_IGEMM_STAGE=0" "MD_IGEMM_STAGE=1"; do
    echo "== $env"; env $env timeout 300 python tools/epi_bench.py 2>&1 | tail -24
  done
  for env in "MD_IGEMM_STAGE=1" "MD_IGEMM_STAGE=0" "MD_IGEMM_STAGE=1" "MD_IGEMM_STAGE=0"; do
    echo "== $env"
    env $env timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
    env $env timeout 300 python bench.py --frames-per-gpu 8 --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  done
  echo "== e2e"; timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x 2>&1 | tail -3
} > gpurun_out/r3j.txt 2>&1
cat gpurun_out/r3j.txt
