#!/bin/bash
# round 3, run T: sanity of the tree after the A/B switches were removed: all GPU tests, smoke, default bench line
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|NCCL\|RCCL\|^$" | tail -3 > gpurun_out/r3t.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 >> gpurun_out/r3t.txt
timeout 900 python bench.py > gpurun_out/r3t_bench_n1.json 2> gpurun_out/r3t_bench_n1.err
grep -o '"value": [0-9.]*' gpurun_out/r3t_bench_n1.json | head -2 >> gpurun_out/r3t.txt
grep -o '"frac": [0-9.]*' gpurun_out/r3t_bench_n1.json >> gpurun_out/r3t.txt
cat gpurun_out/r3t.txt
