#!/bin/bash
# round 2, run AT: KS1 (1x1 / linear) LDS-DMA issue path: igemm tests + e2e subset, launch floor, configs[1] bench
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -m gpu -q -k "igemm or merged_and_forked or (golden and small_b2) or bit_identical" 2>&1 | tail -3 > gpurun_out/r2at.txt
timeout 60 python tools/launch_floor.py 2>&1 | grep "md_igemm M=8192" >> gpurun_out/r2at.txt
timeout 200 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1 >> gpurun_out/r2at.txt
cat gpurun_out/r2at.txt
