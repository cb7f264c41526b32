#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -k "fp8 or igemm_m32 or attention" 2>&1 | tail -15 > gpurun_out/r2k_fp8_tests.log
timeout 400 python bench.py --size 96 --steps 3 --warmup 1 --no-extra --no-roofline --fp8-attention > gpurun_out/r2k_bench_768_fp8.log 2>&1
timeout 400 python bench.py --size 96 --steps 3 --warmup 1 --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2k_bench_768_fp16.log 2>&1
timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline --fp8-attention > gpurun_out/r2k_bench_512_fp8.log 2>&1
cat gpurun_out/r2k_fp8_tests.log | tail -12; grep -h "fp8" gpurun_out/parity_e2e.log
for f in r2k_bench_768_fp8 r2k_bench_768_fp16 r2k_bench_512_fp8; do grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1; done
grep -o '"parity_full_size": {[^}]*}' gpurun_out/r2k_bench_768_fp8.log
