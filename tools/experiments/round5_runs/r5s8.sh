#!/bin/bash
# round 5, run S8/S9: LayerNorm-folded md_igemm repeatability, diagnostic variants of the epilogue transform
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
P=tools/experiments/round5_runs/ln_repeat.py
V=tools/experiments/round5_runs/variants
{ for f in $V/*.so; do timeout 300 python $P $f 40; done; } 2>&1 | grep LNREP | grep -v "NO Layer" | tee gpurun_out/r5s9_ln_repeat.txt
