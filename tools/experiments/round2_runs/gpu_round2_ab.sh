#!/bin/bash
# round 2, run AB: batched epilogue loads: igemm tests, e2e parity, epilogue probe, launch floor, bench (1 and 8 frames)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "igemm or embedding" 2>&1 | tail -4 > gpurun_out/r2ab_igemm_tests.log
timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -x 2>&1 | tail -4 > gpurun_out/r2ab_e2e.log
timeout 300 python tools/igemm_epilogue_probe.py 2>&1 | grep "M=" > gpurun_out/r2ab_epilogue_probe.txt
timeout 200 python tools/launch_floor.py 2>&1 | grep "per launch" > gpurun_out/r2ab_launch_floor.txt
timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2ab_bench.log 2>&1
timeout 300 python bench.py --frames-per-gpu 8 --steps 3 --warmup 1 --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2ab_bench_f8.log 2>&1
tail -2 gpurun_out/r2ab_igemm_tests.log; tail -2 gpurun_out/r2ab_e2e.log; cat gpurun_out/r2ab_epilogue_probe.txt gpurun_out/r2ab_launch_floor.txt
for f in r2ab_bench r2ab_bench_f8; do grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1; done
