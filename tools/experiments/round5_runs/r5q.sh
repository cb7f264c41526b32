#!/bin/bash
# round 5, run Q: GroupNorm finalize folded into the apply launch for batches <= 4 (gn_apply_parts): parity, bench A/B (MD_GN_FUSED=0 | 1),
# end-to-end parity tests
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x --timeout 600 -k "groupnorm" 2>&1 | tail -4 | tee gpurun_out/r5q_tests.txt
for i in 1 2 3; do for v in 0 1; do
  MD_GN_FUSED=$v timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('MD_GN_FUSED=$v', 'frames/s', round(d['value'],4), 'ms', round(d['ms_per_step'],1))"
done; done 2>&1 | tee gpurun_out/r5q_bench_ab.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_vae.py -q --timeout 900 2>&1 | tail -4 | tee -a gpurun_out/r5q_tests.txt
grep -E "final latent|worst ratio" gpurun_out/parity_fullsize.log | tail -5
