#!/bin/bash
# round 4, run C: ring entries in the tuned table -> end-to-end parity (e2e + full-size goldens) and the bench line
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -m gpu -q -x --timeout 600 --timeout-method=thread 2>&1 | tail -6 > gpurun_out/r4c_e2e_tests.txt
cat gpurun_out/r4c_e2e_tests.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r4c_bench.json 2> gpurun_out/r4c_bench.err
tail -3 gpurun_out/r4c_bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r4c_bench.json").read().strip().split("\n")[-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "roofline", d["roofline"]["frac"], d["roofline"]["ms"], "attn", d["roofline_attention"]["achieved"], d["roofline_attention"]["ms"])
print("families", {k: (round(v["step_ms"], 3), v["launches"]) for k, v in d["families_ms_per_batch"].items()})
e = d["extra"]["configs[2]"]
print("configs[2]", e["value"], e["roofline"]["frac"], e["roofline_attention"]["frac"], {k: round(v["step_ms"], 3) for k, v in e["families_ms_per_batch"].items()})
print("gpu_state", json.dumps(d["gpu_state"])[:600])
print("unet", d["unet_ms_per_step"])
PY
