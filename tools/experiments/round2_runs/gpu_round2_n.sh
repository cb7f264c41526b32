#!/bin/bash
# round 2, run N: A/B of the merged pass with tuned 3F shapes
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2n_bench_merged.log 2>&1
MD_MERGE_POSE=0 timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2n_bench_separate.log 2>&1
timeout 300 python bench.py --frames-per-gpu 8 --steps 3 --warmup 1 --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2n_bench_merged_f8.log 2>&1
MD_MERGE_POSE=0 timeout 300 python bench.py --frames-per-gpu 8 --steps 3 --warmup 1 --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r2n_bench_separate_f8.log 2>&1
for f in r2n_bench_merged r2n_bench_separate r2n_bench_merged_f8 r2n_bench_separate_f8; do echo $f; grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1; grep -o '"unet_ms_per_step": {[^}]*}[^}]*}' gpurun_out/$f.log | head -1; done
