#!/bin/bash
# round 5, run R: GroupNorm inside the split-K reduction (md_igemm_params.gn, ABI v10) + gamma/beta loads of gn_small hoisted:
# bit-exactness tests, groupnorm kernel tests, same-box bench A/B (MD_GN_NEXT=0 | 1), step breakdown
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_igemm_gn.py -q --timeout 600 2>&1 | tail -40 | tee gpurun_out/r5r_tests.txt
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x --timeout 600 -k "groupnorm" 2>&1 | tail -4 | tee -a gpurun_out/r5r_tests.txt
for i in 1 2 3; do for v in 0 1; do
  MD_GN_NEXT=$v timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('MD_GN_NEXT=$v', 'frames/s', round(d['value'],4), 'ms', round(d['ms_per_step'],1))"
done; done 2>&1 | tee gpurun_out/r5r_bench_ab.txt
for v in 0 1; do
  MD_GN_NEXT=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('MD_GN_NEXT=$v', 'frames/s', round(d['value'],4), 'configs[2]', round(d['extra']['configs[2]']['value'],4))"
done 2>&1 | tee -a gpurun_out/r5r_bench_ab.txt
timeout 300 python tools/step_breakdown.py 1 2>&1 | head -70 > gpurun_out/r5r_step_breakdown_1frame.txt
head -3 gpurun_out/r5r_step_breakdown_1frame.txt
