#!/bin/bash
# round 2, run AG: full re-tune (all shapes of the table pass + one step, frames 1 and 8) with the final kernels
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/tune_igemm.py gpurun_out/igemm_tuned_full.inc 1 8 > gpurun_out/r2ag_tune.log 2>&1
tail -2 gpurun_out/r2ag_tune.log; wc -l gpurun_out/igemm_tuned_full.inc
