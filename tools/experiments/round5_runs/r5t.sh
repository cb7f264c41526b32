#!/bin/bash
# round 5, run T: the library without packed-fp32 VALU code in the GEMM translation units: repeatability (whole sampler, every launch of a
# step), the fused-GroupNorm tests incl. the bit-identical sampler check, same-box bench A/B against the packed build
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
OLD=$PWD/tools/experiments/round5_runs/variants/libmd_packed_fp32.so
{
echo "== repeat_check, new build"; timeout 300 python tools/repeat_check.py 8 1 2>&1 | tail -3; timeout 400 python tools/repeat_check.py 4 8 2>&1 | tail -3
echo "== repeat_check, packed-fp32 build"; MD_HIP_LIB=$OLD timeout 300 python tools/repeat_check.py 4 1 2>&1 | tail -2
echo "== per-launch probe, new build"; timeout 400 python tools/call_repeat_probe.py 1 8 2>&1 | cut -c1-300 | tail -4; timeout 400 python tools/call_repeat_probe.py 8 4 2>&1 | cut -c1-300 | tail -4
echo "== LayerNorm-folded GEMMs, 60 runs each"; timeout 300 python tools/experiments/round5_runs/ln_repeat.py - 60 2>&1 | grep LNREP
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5t_repeat.txt
timeout 900 python -m pytest tests/test_gpu_igemm_gn.py -q --timeout 600 --tb=line 2>&1 | grep -E "differs|declined|Error|passed|failed" | cut -c1-300 | tail -8 | tee gpurun_out/r5t_tests.txt
for i in 1 2 3; do for v in old new; do
  L=""; if [ $v = old ]; then L=$OLD; fi
  MD_HIP_LIB=$L timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v build', 'frames/s', round(d['value'],4), 'configs[2]', round(d['extra']['configs[2]']['value'],4))"
done; done 2>&1 | tee gpurun_out/r5t_bench_ab.txt
