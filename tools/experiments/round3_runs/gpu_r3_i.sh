#!/bin/bash
# round 3, run I: forked skip-conv / zero-conv branches + k-group tuned table for the 64x64-level convs: tests, same-box A/B
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== e2e"; timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x 2>&1 | tail -3
  echo "== fullsize"; timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "baseline or batch8" 2>&1 | tail -3
  for env in "MD_FORK=1" "MD_FORK=0" "MD_FORK=1" "MD_FORK=0"; do
    echo "== $env"
    env $env timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
    env $env timeout 300 python bench.py --frames-per-gpu 8 --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  done
} > gpurun_out/r3i.txt 2>&1
cat gpurun_out/r3i.txt
