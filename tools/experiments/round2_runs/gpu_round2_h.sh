#!/bin/bash
# GPU box: round-2 run H -- default bench line (all legs incl. cpu_baseline), smoke, 768^2 line, sequence line, step breakdowns
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/r2h_bench_default.log 2>&1
timeout 200 python __graft_entry__.py smoke > gpurun_out/r2h_smoke.log 2>&1
timeout 400 python bench.py --size 96 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > gpurun_out/r2h_bench_768.log 2>&1
timeout 400 python bench.py --sequence 16 --frames-per-gpu 8 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > gpurun_out/r2h_bench_seq16.log 2>&1
timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r2h_step_breakdown_1frame.txt 2>&1
timeout 300 python tools/step_breakdown.py 8 > gpurun_out/r2h_step_breakdown_8frames.txt 2>&1
tail -c 2500 gpurun_out/r2h_bench_default.log; tail -3 gpurun_out/r2h_smoke.log
for f in r2h_bench_768 r2h_bench_seq16; do grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1; done
head -3 gpurun_out/r2h_step_breakdown_8frames.txt | tail -2
