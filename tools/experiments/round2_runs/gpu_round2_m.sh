#!/bin/bash
# round 2, run M: tune the merged-pass shapes (3F samples), F = 1 and 8
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
TUNE_ONLY_NEW=1 timeout 900 python tools/tune_igemm.py gpurun_out/igemm_tuned_merged.inc 1 8 > gpurun_out/r2m_tune.log 2>&1
tail -5 gpurun_out/r2m_tune.log; wc -l gpurun_out/igemm_tuned_merged.inc
