#!/bin/bash
# round 3, run K: staged GEGLU epilogue: tests + A/B; linear-layer re-tune with the residual stream attached
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== kernels"; timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "igemm or linear or geglu" 2>&1 | tail -3
  for env in "MD_IGEMM_STAGE=0" "MD_IGEMM_STAGE=1"; do
    echo "== $env"; env $env timeout 300 python tools/epi_bench.py 2>&1 | grep geglu
  done
  for env in "MD_IGEMM_STAGE=1" "MD_IGEMM_STAGE=0"; do
    echo "== $env"
    env $env timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
    env $env timeout 300 python bench.py --frames-per-gpu 8 --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  done
  echo "== tune linear (residual stream attached)"
  TUNE_FILTER=linear TUNE_RES=1 timeout 900 python tools/tune_igemm.py gpurun_out/tuned_linear_r3k.inc 1 8 2>&1 | tail -150
} > gpurun_out/r3k.txt 2>&1
tail -c 30000 gpurun_out/r3k.txt
