#!/bin/bash
# round 3, run U: full re-tune (batches of 1 and 8 frames) on the final kernels: tiled weights, residual stream attached to the linears
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
TUNE_RES=1 timeout 1200 python tools/tune_igemm.py gpurun_out/tuned_all_r3u.inc 1 8 > gpurun_out/r3u.txt 2>&1
tail -3 gpurun_out/r3u.txt | cut -c1-200
