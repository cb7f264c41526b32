#!/bin/bash
# round 5, run E: md_ff_block inside the step: per-launch breakdowns and bench A/B (MD_FF_BLOCK=0 | 320), e2e / full-size parity tests
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 320; do
  MD_FF_BLOCK=$v timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r5e_step_1f_ff$v.txt 2>&1; head -3 gpurun_out/r5e_step_1f_ff$v.txt | tail -2
  MD_FF_BLOCK=$v timeout 300 python tools/step_breakdown.py 8 > gpurun_out/r5e_step_8f_ff$v.txt 2>&1; head -3 gpurun_out/r5e_step_8f_ff$v.txt | tail -2
done
for v in 0 320 0 320; do
  MD_FF_BLOCK=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/r5e_bench_err_$v.txt | tail -1 > gpurun_out/r5e_bench_ff$v.json
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r5e_bench_ff{v}.json"))
    e = d.get("extra", {}).get("configs[2]", {})
    print("MD_FF_BLOCK=" + v, "frames/s", round(d["value"], 4), "ms", round(d["ms_per_step"], 1), "configs[2]", round(e.get("value", 0), 3))
except Exception as ex:
    print("MD_FF_BLOCK=" + v, "bench failed", ex)
PY
done 2>&1 | tee gpurun_out/r5e_bench_ab.txt
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py tests/test_gpu_ffblock.py -q --timeout 900 2>&1 | tail -6 | tee gpurun_out/r5e_tests.txt
