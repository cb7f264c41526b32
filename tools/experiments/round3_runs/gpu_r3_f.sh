#!/bin/bash
# round 3, run F: the default bench line (what the driver runs) + the other quoted workloads, full-size parity log
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_fullsize.log
timeout 900 python bench.py > gpurun_out/r3f_bench_n1.json 2> gpurun_out/r3f_bench_n1.err
timeout 300 python bench.py --size 96 --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r3f_bench_n1_768x768.json 2>/dev/null
timeout 300 python bench.py --size 96 --fp8-attention --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r3f_bench_n1_768x768_fp8.json 2>/dev/null
timeout 300 python bench.py --sequence 16 --frames-per-gpu 8 --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r3f_bench_n1_seq16.json 2>/dev/null
timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q 2>&1 | tail -3 > gpurun_out/r3f.txt
for f in gpurun_out/r3f_bench_*.json; do echo "$f: $(grep -o '"value": [0-9.]*' $f | head -2 | tr '\n' ' ')" >> gpurun_out/r3f.txt; done
grep -o '"frac": [0-9.]*' gpurun_out/r3f_bench_n1.json >> gpurun_out/r3f.txt
cat gpurun_out/r3f.txt
