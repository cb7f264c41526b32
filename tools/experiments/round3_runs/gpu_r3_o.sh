#!/bin/bash
# round 3, run O: two-source issue path (KS = 4) validation + 3x3 re-tune with the new issue paths
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== kernels"; timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "igemm" 2>&1 | tail -3
  echo "== e2e"; timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x 2>&1 | tail -3
  echo "== bench"
  timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  timeout 300 python bench.py --frames-per-gpu 8 --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  echo "== tune conv3"
  TUNE_FILTER=conv3 timeout 1200 python tools/tune_igemm.py gpurun_out/tuned_conv3_r3o.inc 1 8 2>&1 | tail -120
} > gpurun_out/r3o.txt 2>&1
tail -c 40000 gpurun_out/r3o.txt
