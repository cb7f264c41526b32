#!/bin/bash
# GPU box: attention v3 (software-pipelined) vs v2: correctness (kernel tests under both), microbench, one bench line each
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
MD_ATTN_V=3 timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attention" 2>&1 | tail -8 > gpurun_out/r2b_attn_tests_v3.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q -k "attention" 2>&1 | tail -4 > gpurun_out/r2b_attn_tests_v2.log
timeout 200 python tools/attn_bench.py > gpurun_out/r2b_attn_bench_v2.txt 2>&1
MD_ATTN_V=3 timeout 200 python tools/attn_bench.py > gpurun_out/r2b_attn_bench_v3.txt 2>&1
MD_ATTN_V=3 MD_ATTN_QF=1 timeout 200 python tools/attn_bench.py > gpurun_out/r2b_attn_bench_v3_qf1.txt 2>&1
MD_ATTN_V=3 timeout 300 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -k "golden and small_b1" 2>&1 | tail -4 > gpurun_out/r2b_e2e_v3.log
MD_ATTN_V=3 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extra > gpurun_out/r2b_bench_v3.log 2>&1
MD_ATTN_V=3 timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extra --frames-per-gpu 8 --steps 2 --warmup 1 > gpurun_out/r2b_bench_v3_f8.log 2>&1
timeout 300 python bench.py --no-cpu-baseline --no-roofline --no-extra --frames-per-gpu 8 --steps 2 --warmup 1 > gpurun_out/r2b_bench_v2_f8.log 2>&1
cat gpurun_out/r2b_attn_tests_v3.log; echo; paste -d'|' gpurun_out/r2b_attn_bench_v2.txt gpurun_out/r2b_attn_bench_v3.txt | cut -c1-170; cat gpurun_out/r2b_attn_bench_v3_qf1.txt
for f in r2b_bench_v3 r2b_bench_v3_f8 r2b_bench_v2_f8; do grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1; done
