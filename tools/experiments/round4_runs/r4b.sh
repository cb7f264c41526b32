#!/bin/bash
# round 4, run B: ring tuner on the mid-size layers (1024 < M <= 16384: the 32x32 / 64x64 levels of a one-frame step)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python tools/tune_ring.py gpurun_out/r4b_tuned.inc --mmin 1025 --mmax 16384 > gpurun_out/r4b_tune_ring.txt 2>&1
tail -90 gpurun_out/r4b_tune_ring.txt
