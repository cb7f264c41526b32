#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -30 > gpurun_out/r2e_pytest.log
timeout 200 python tools/attn_bench.py > gpurun_out/r2e_attn_bench_v3.txt 2>&1
MD_ATTN_V=2 timeout 200 python tools/attn_bench.py > gpurun_out/r2e_attn_bench_v2.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extra > gpurun_out/r2e_bench.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extra --frames-per-gpu 8 --steps 2 --warmup 1 > gpurun_out/r2e_bench_f8.log 2>&1
tail -4 gpurun_out/r2e_pytest.log
paste -d'|' gpurun_out/r2e_attn_bench_v2.txt gpurun_out/r2e_attn_bench_v3.txt | grep -v amdgpu | cut -c1-160
for f in r2e_bench r2e_bench_f8; do grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1; done
