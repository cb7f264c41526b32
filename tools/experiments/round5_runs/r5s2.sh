#!/bin/bash
# round 5, run S2: where does conv(gn_next=) change the sampler's result?
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/experiments/round5_runs/gn_next_bisect.py 2>&1 | grep -E "_GN_NEXT|Error|error" | tee gpurun_out/r5s2_bisect.txt
MD_GN_REDUCE=0 timeout 600 python tools/experiments/round5_runs/gn_next_bisect.py 2>&1 | grep -E "_GN_NEXT|Error|error" | tee -a gpurun_out/r5s2_bisect.txt
