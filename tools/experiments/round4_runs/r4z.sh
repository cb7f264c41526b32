#!/bin/bash
# round 4, run Z: the counters of run Y again with the weights TILED as the engine stores them (run Y timed row-major weights)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out/igpmc_r4z
export TMPDIR=/tmp
R=$(pwd); D=$R/gpurun_out/igpmc_r4z
{
  (cd /tmp && timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-include-regex igemm -d "$D" -o sq --output-format csv -- python $R/tools/igemm_pmc.py > "$D/sq.log" 2>&1; echo sq rc=$?)
  (cd /tmp && timeout 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-include-regex igemm -d "$D" -o l2 --output-format csv -- python $R/tools/igemm_pmc.py > "$D/l2.log" 2>&1; echo l2 rc=$?)
  python - "$D" <<'PY'
import csv, sys, collections, os
d = sys.argv[1]
for tag in ("sq", "l2"):
    p = os.path.join(d, f"{tag}_counter_collection.csv")
    if not os.path.exists(p):
        print(tag, "missing"); continue
    rows = list(csv.DictReader(open(p)))
    byk = collections.OrderedDict()
    for r in rows:
        if "igemm" not in r["Kernel_Name"] or "reduce" in r["Kernel_Name"]:
            continue
        key = (r["Dispatch_Id"], r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("mdig::", "")[:70], r["Grid_Size"])
        byk.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for k, v in byk.items():
        print(tag, k[0], k[1], "grid", k[2], " ".join(f"{a}={b:.6g}" for a, b in v.items()))
PY
  rm -f "$D"/*counter_collection.csv
} > gpurun_out/r4z_counters.txt 2>&1
cat gpurun_out/r4z_counters.txt | tail -3
timeout 600 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | tail -1 > gpurun_out/r4z_bench.json; python -c "
import json; d=json.load(open('gpurun_out/r4z_bench.json')); print('bench', round(d['value'],4), 'frames/s', round(d['ms_per_step'],1), 'ms; igemm frac', round(d['roofline']['frac'],4), 'sclk after', d['gpu_state']['after_timed_region']['card0']['sclk clock speed:'])" | tee gpurun_out/r4z_bench.txt
