#!/bin/bash
# round 4, run X: the folded-LayerNorm row statistics shared between the n-waves of a wave row -- parity of every LN path, then the cost probe again
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_igemm_ring.py -m gpu -q -k "layernorm or ln or ring or geglu" --timeout 600 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee gpurun_out/r4x_ln_tests.txt
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_clip.py -m gpu -q -x --timeout 600 2>&1 | grep -E "passed|failed|error|Error" | tail -5 | tee gpurun_out/r4x_e2e_tests.txt
timeout 300 python tools/ln_cost_probe.py 2>&1 | grep "M=" | tee gpurun_out/r4x_ln_cost.txt
timeout 600 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > gpurun_out/r4x_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r4x_bench.json')); print('bench', round(d['value'],4), 'frames/s', round(d['ms_per_step'],1), 'ms; igemm frac', round(d['roofline']['frac'],4))" | tee gpurun_out/r4x_bench.txt
