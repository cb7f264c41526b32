#!/bin/bash
# round 5, run A: first GPU contact of md_ff_block -- parity tests, microbenchmark against the md_igemm launches it replaces,
# per-launch step breakdowns with and without it (1 and 8 frames), same-box bench A/B
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ffblock.py -q -x --timeout 300 2>&1 | tail -15 | tee gpurun_out/r5a_ffblock_tests.txt
timeout 300 python tools/ffblock_bench.py > gpurun_out/r5a_ffblock_bench.txt 2>&1; tail -12 gpurun_out/r5a_ffblock_bench.txt
for v in 0 320 320,640; do
  MD_FF_BLOCK=$v timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r5a_step_1f_ff$v.txt 2>&1; head -2 gpurun_out/r5a_step_1f_ff$v.txt
  MD_FF_BLOCK=$v timeout 300 python tools/step_breakdown.py 8 > gpurun_out/r5a_step_8f_ff$v.txt 2>&1; head -2 gpurun_out/r5a_step_8f_ff$v.txt
done
for v in 0 320 320,640; do
  MD_FF_BLOCK=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>gpurun_out/r5a_bench_err_$v.txt | tail -1 > gpurun_out/r5a_bench_ff$v.json
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/r5a_bench_ff{v}.json"))
    e = d.get("extra", {}).get("configs[2]", {})
    print("MD_FF_BLOCK=" + v, "frames/s", round(d["value"], 4), "ms", round(d["ms_per_step"], 1), "configs[2]", round(e.get("value", 0), 3))
except Exception as ex:
    print("MD_FF_BLOCK=" + v, "bench failed", ex)
PY
done 2>&1 | tee gpurun_out/r5a_bench_ab.txt
