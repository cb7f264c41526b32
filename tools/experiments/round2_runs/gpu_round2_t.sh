#!/bin/bash
# round 2, run T: full GPU tier + default bench after the merged pass / igemm loop changes
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r2t_pytest.log
timeout 200 python __graft_entry__.py smoke > gpurun_out/r2t_smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/r2t_bench.log 2>&1
tail -5 gpurun_out/r2t_pytest.log; tail -2 gpurun_out/r2t_smoke.log; tail -c 1200 gpurun_out/r2t_bench.log
