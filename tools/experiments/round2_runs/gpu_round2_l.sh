#!/bin/bash
# round 2, run L: second parameter set kernels + merged pose/UNet encoder pass: kernel tests, e2e parity, A/B bench
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "second_parameter" 2>&1 | tail -15 > gpurun_out/r2l_kernel_tests.log
timeout 700 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r2l_e2e_tests.log
timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2l_bench_merged.log 2>&1
MD_MERGE_POSE=0 timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2l_bench_separate.log 2>&1
timeout 300 python bench.py --frames-per-gpu 8 --steps 3 --warmup 1 --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2l_bench_merged_f8.log 2>&1
MD_MERGE_POSE=0 timeout 300 python bench.py --frames-per-gpu 8 --steps 3 --warmup 1 --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2l_bench_separate_f8.log 2>&1
cat gpurun_out/r2l_kernel_tests.log | tail -8; cat gpurun_out/r2l_e2e_tests.log | tail -12
for f in r2l_bench_merged r2l_bench_separate r2l_bench_merged_f8 r2l_bench_separate_f8; do echo $f; grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1; grep -o '"launches_per_step": [0-9]*' gpurun_out/$f.log | head -1; tail -2 gpurun_out/$f.log | cut -c1-300; done
