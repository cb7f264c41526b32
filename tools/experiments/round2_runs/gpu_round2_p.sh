#!/bin/bash
# round 2, run P: compute_tile with all fragment reads up front + DMA issue inside: igemm tests, per-shape roofline, bench
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "igemm" 2>&1 | tail -6 > gpurun_out/r2p_igemm_tests.log
timeout 600 python tools/igemm_shape_roofline.py > gpurun_out/r2p_igemm_shape_roofline.txt 2>&1
timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2p_bench.log 2>&1
timeout 300 python bench.py --frames-per-gpu 8 --steps 3 --warmup 1 --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2p_bench_f8.log 2>&1
tail -4 gpurun_out/r2p_igemm_tests.log; head -12 gpurun_out/r2p_igemm_shape_roofline.txt
for f in r2p_bench r2p_bench_f8; do grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1; done
