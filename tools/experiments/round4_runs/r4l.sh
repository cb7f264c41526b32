#!/bin/bash
# round 4, run L: envelope tests + attention / e2e tests with the round-4 attention loop; short bench (1 and 8 frames per batch)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "envelope" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -q -x -k "attention or golden or spike" 2>&1 | tail -4
cat gpurun_out/parity_fullsize.log | grep envelope
for fpg in 1 8; do
timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 4 --warmup 2 --frames-per-gpu $fpg 2>/dev/null | tail -1 > gpurun_out/r4l_bench_fpg$fpg.json
python - <<PY
import json
d=json.loads(open("gpurun_out/r4l_bench_fpg$fpg.json").read())
print("fpg$fpg", round(d["value"],4), "frames/s", round(d["ms_per_step"],1), "ms/batch; igemm frac", round(d["roofline"]["frac"],4), "attention", {k:(round(v,4) if isinstance(v,float) else v) for k,v in d["roofline_attention"].items() if k in ("achieved","frac","ms","mfma_busy")})
print({k:round(v["ms"],1) for k,v in d["families_ms_per_batch"].items()})
PY
done
