#!/bin/bash
# round 3, run B: kernel + parity tests of the k-group / GroupNorm-partials build (epilogue without the register-resident partials),
# 1-rank RCCL test, same-box A/B of the two features, per-shape step breakdown, tuner pass over the small-M shapes with k-groups.
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== kernels"; timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x 2>&1 | tail -4
  echo "== rccl"; timeout 600 python -m pytest tests/test_gpu_rccl.py -m gpu -q -x 2>&1 | tail -12
  echo "== fullsize"; timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -12
  for env in "MD_IGEMM_KG=1 MD_GN_FUSE=1" "MD_IGEMM_KG=0 MD_GN_FUSE=1" "MD_IGEMM_KG=1 MD_GN_FUSE=0" "MD_IGEMM_KG=0 MD_GN_FUSE=0" "MD_IGEMM_KG=1 MD_GN_FUSE=1"; do
    echo "== $env"
    env $env timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
    env $env timeout 300 python bench.py --frames-per-gpu 8 --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  done
} > gpurun_out/r3b.txt 2>&1
timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r3b_step_breakdown_1frame.txt 2>&1
MD_IGEMM_KG=0 timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r3b_step_breakdown_1frame_kg0.txt 2>&1
TUNE_FILTER=smallm timeout 900 python tools/tune_igemm.py gpurun_out/r3b_tuned_smallm.inc 1 > gpurun_out/r3b_tune.log 2>&1
tail -3 gpurun_out/r3b_tune.log >> gpurun_out/r3b.txt
cat gpurun_out/r3b.txt
