#!/bin/bash
# round 3, run H: tuner pass with k-groups over the mid-M shapes (64x64 level of a one-frame step, low levels of 8-frame batches)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
TUNE_FILTER=midm timeout 1500 python tools/tune_igemm.py gpurun_out/r3h_tuned_midm.inc 1 8 > gpurun_out/r3h_tune.log 2>&1
tail -4 gpurun_out/r3h_tune.log
