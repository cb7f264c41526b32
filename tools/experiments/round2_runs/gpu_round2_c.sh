#!/bin/bash
# GPU box: attention v3 with XCD-aware placement + deep prefetch ring: tests, microbench A/B, bench lines
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
MD_ATTN_V=3 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attention" 2>&1 | tail -8 > gpurun_out/r2c_attn_tests_v3.log
MD_ATTN_V=3 MD_ATTN_P=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attention" 2>&1 | tail -4 > gpurun_out/r2c_attn_tests_v3_p1.log
timeout 200 python tools/attn_bench.py > gpurun_out/r2c_attn_bench_v2.txt 2>&1
MD_ATTN_V=3 MD_ATTN_P=1 timeout 200 python tools/attn_bench.py > gpurun_out/r2c_attn_bench_v3_p1.txt 2>&1
MD_ATTN_V=3 timeout 200 python tools/attn_bench.py > gpurun_out/r2c_attn_bench_v3.txt 2>&1
MD_ATTN_V=3 MD_ATTN_QF=1 timeout 200 python tools/attn_bench.py > gpurun_out/r2c_attn_bench_v3_qf1.txt 2>&1
MD_ATTN_V=3 timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r2c_e2e_v3.log
MD_ATTN_V=3 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extra > gpurun_out/r2c_bench_v3.log 2>&1
MD_ATTN_V=3 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extra --frames-per-gpu 8 --steps 2 --warmup 1 > gpurun_out/r2c_bench_v3_f8.log 2>&1
cat gpurun_out/r2c_attn_tests_v3.log gpurun_out/r2c_attn_tests_v3_p1.log gpurun_out/r2c_e2e_v3.log; echo
paste -d'|' gpurun_out/r2c_attn_bench_v2.txt gpurun_out/r2c_attn_bench_v3_p1.txt gpurun_out/r2c_attn_bench_v3.txt gpurun_out/r2c_attn_bench_v3_qf1.txt | sed 's/B=\([0-9]\) nq=\([0-9]*\) n0=\([0-9]*\) n1=\([0-9]*\) d=\([0-9]*\)://g' | cut -c1-200
for f in r2c_bench_v3 r2c_bench_v3_f8; do grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1; done
