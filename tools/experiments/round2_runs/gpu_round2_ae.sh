#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "geglu or m32 or folded" 2>&1 | tail -3
timeout 300 python tools/igemm_epilogue_probe.py 2>&1 | grep "act=2" 
timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
timeout 300 python bench.py --frames-per-gpu 8 --steps 3 --warmup 1 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
MD_MERGE_POSE=0 timeout 300 python bench.py --frames-per-gpu 8 --steps 3 --warmup 1 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
