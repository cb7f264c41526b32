#!/bin/bash
# round 3, run A: correctness of the k-group / GroupNorm-partials kernels, then same-box A/B of the two features end to end.
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== kernels"; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -15
  echo "== fullsize + e2e"; timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_e2e.py -m gpu -q -x 2>&1 | tail -8
  for env in "MD_IGEMM_KG=1 MD_GN_FUSE=1" "MD_IGEMM_KG=0 MD_GN_FUSE=1" "MD_IGEMM_KG=1 MD_GN_FUSE=0" "MD_IGEMM_KG=0 MD_GN_FUSE=0" "MD_IGEMM_KG=1 MD_GN_FUSE=1"; do
    echo "== $env"
    env $env timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
    env $env timeout 300 python bench.py --frames-per-gpu 8 --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  done
} > gpurun_out/r3a.txt 2>&1
cat gpurun_out/r3a.txt
