#!/bin/bash
# round 3, run L: tiled weight storage, isolated cold-weight A/B (bit-identity asserted inside)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
{
  timeout 600 python tools/wtile_bench.py 2>&1 | tail -30
} > gpurun_out/r3l.txt 2>&1
cat gpurun_out/r3l.txt
