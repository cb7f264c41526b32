#!/bin/bash
# round 2, run AS: default bench of the final tree (refresh profiles/round2_bench_n1.json)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 220 python bench.py > gpurun_out/r2as_bench.log 2>&1
grep -o '"value": [0-9.]*' gpurun_out/r2as_bench.log | head -3
