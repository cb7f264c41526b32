#!/bin/bash
# GPU box: round-2 validation run A (parity + bench + step breakdown).  Writes everything under gpurun_out/.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2a_pytest.log
MD_RES_LO=0 timeout 400 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "c1_b1" 2>&1 | tail -5 > gpurun_out/r2a_pytest_single_term.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2a_bench.log 2>&1
MD_RES_LO=0 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extra > gpurun_out/r2a_bench_single_term.log 2>&1
timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r2a_step_breakdown_1frame.txt 2>&1
tail -3 gpurun_out/r2a_pytest.log; tail -c 1500 gpurun_out/r2a_bench.log; tail -c 400 gpurun_out/r2a_bench_single_term.log
