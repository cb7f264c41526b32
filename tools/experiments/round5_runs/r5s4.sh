#!/bin/bash
# round 5, run S4: which component makes the full-width sampler unrepeatable?
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
P=tools/experiments/round5_runs/repeat_probe2.py
{
timeout 300 python $P 4 2>&1 | grep PROBE
timeout 300 python $P 10 2>&1 | grep PROBE
timeout 300 python $P 4 nograph 2>&1 | grep PROBE
MD_FF_BLOCK=0 timeout 300 python $P 4 2>&1 | grep PROBE
MD_IGEMM_TUNED=0 timeout 300 python $P 4 2>&1 | grep PROBE
MD_MERGE_POSE=0 timeout 300 python $P 4 2>&1 | grep PROBE
MD_GN_FUSE=0 MD_GN_NEXT=0 timeout 300 python $P 4 2>&1 | grep PROBE
MD_FF_BLOCK=0 MD_IGEMM_TUNED=0 MD_GN_NEXT=0 timeout 300 python $P 4 2>&1 | grep PROBE
} | tee gpurun_out/r5s4_repeat.txt
