#!/bin/bash
# round 4, run AC: bench.py's multi-GPU branches on a 1-rank RCCL group (--force-sharded; fixed per-GPU work at every N, the configs[3] shape as
# `extra` at N > 1), the RCCL test file, and the default N = 1 line again
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_rccl.py -m gpu -q -x --timeout 900 2>&1 | grep -vE "amdgpu.ids|^$" | tail -15 | tee gpurun_out/r4ac_rccl_tests.txt
timeout 600 python bench.py --gpus 1 --force-sharded --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/r4ac_sharded_err.txt | tail -1 > gpurun_out/r4ac_bench_sharded.json
python -c "
import json; d=json.load(open('gpurun_out/r4ac_bench_sharded.json')); print('1-rank sharded: value', round(d['value'],4), 'ms', round(d['ms_per_step'],1), '| extra', {k:(round(v['value'],3), round(v['ms_per_step'],1)) for k,v in d['extra'].items()}, '|', d['config']['workload'][:60])" | tee gpurun_out/r4ac_bench_sharded.txt
tail -5 gpurun_out/r4ac_sharded_err.txt
timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r4ac_bench_n1.json
python -c "
import json; d=json.load(open('gpurun_out/r4ac_bench_n1.json')); print('N=1 default: value', round(d['value'],4), 'ms', round(d['ms_per_step'],1), 'frac', round(d['roofline']['frac'],4), '| extra', {k:round(v['value'],3) for k,v in d['extra'].items()})" | tee gpurun_out/r4ac_bench_n1.txt
