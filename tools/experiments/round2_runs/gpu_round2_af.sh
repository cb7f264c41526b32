#!/bin/bash
# round 2, run AF: re-tune the 1x1 / linear shapes (epilogue-dominated) after the epilogue changes, frames 1 and 8
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
TUNE_FILTER=linear timeout 1100 python tools/tune_igemm.py gpurun_out/igemm_tuned_linear.inc 1 8 > gpurun_out/r2af_tune.log 2>&1
tail -3 gpurun_out/r2af_tune.log; wc -l gpurun_out/igemm_tuned_linear.inc
