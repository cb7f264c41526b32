#!/bin/bash
# round 4, run J: pipe_bench with the packed-fp16 polynomial exp2 roles; attention loop with polynomial exp2 (accuracy + timing)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 tools/bin/pipe_bench > gpurun_out/r4j_pipe_bench.txt 2>&1; echo "pipe rc=$?"
timeout 600 python tools/attn_ab2.py > gpurun_out/r4j_attn_poly.txt 2>&1; echo "ab2 rc=$?"
tail -n +18 gpurun_out/r4j_pipe_bench.txt; cat gpurun_out/r4j_attn_poly.txt
