#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python tools/launch_floor.py 2>&1 | grep "per launch" > gpurun_out/r2z_launch_floor.txt
timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -q -k "merged_and_forked" 2>&1 | tail -3
cat gpurun_out/r2z_launch_floor.txt
