#!/bin/bash
# round 5, run S3: repeatability probe of the sampler on one model object
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python tools/experiments/round5_runs/repeat_probe.py 2>&1 | grep -E "equal|first|Error|error" | tee gpurun_out/r5s3_repeat.txt
