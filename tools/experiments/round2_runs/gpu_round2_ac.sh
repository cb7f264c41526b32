#!/bin/bash
# round 2, run AC: batched epilogue loads (no array select): igemm tests, epilogue probe, launch floor, bench
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "igemm" 2>&1 | tail -3 > gpurun_out/r2ac_igemm_tests.log
timeout 300 python tools/igemm_epilogue_probe.py 2>&1 | grep "M=" > gpurun_out/r2ac_epilogue_probe.txt
timeout 200 python tools/launch_floor.py 2>&1 | grep "per launch" > gpurun_out/r2ac_launch_floor.txt
timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2ac_bench.log 2>&1
timeout 300 python bench.py --frames-per-gpu 8 --steps 3 --warmup 1 --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2ac_bench_f8.log 2>&1
tail -2 gpurun_out/r2ac_igemm_tests.log; cat gpurun_out/r2ac_epilogue_probe.txt gpurun_out/r2ac_launch_floor.txt
for f in r2ac_bench r2ac_bench_f8; do grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1; done
