#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
TUNE_FILTER=shortk timeout 900 python tools/tune_igemm.py gpurun_out/igemm_tuned_shortk.inc 8 > gpurun_out/r2aj_tune.log 2>&1
grep "^    {" gpurun_out/r2aj_tune.log | cut -c1-200
