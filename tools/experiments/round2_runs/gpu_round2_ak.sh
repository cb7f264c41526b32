#!/bin/bash
# round 2, run AK: validation of the final igemm epilogue (restrict, single column): full GPU tier + default bench
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/r2ak_pytest.log
timeout 200 python __graft_entry__.py smoke > gpurun_out/r2ak_smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/r2ak_bench.log 2>&1
tail -2 gpurun_out/r2ak_pytest.log; tail -2 gpurun_out/r2ak_smoke.log; grep -o '"value": [0-9.]*' gpurun_out/r2ak_bench.log | head -3
