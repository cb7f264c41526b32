#!/bin/bash
# round 4, run Y: SQ / L2 counters of isolated md_igemm shapes (2-stage and ring forms), then validation of the final tree:
# whole GPU test tier, smoke, the default bench line
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out/igpmc_r4y
export TMPDIR=/tmp
R=$(pwd); D=$R/gpurun_out/igpmc_r4y
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
} > gpurun_out/r4y_counters.txt 2>&1
tail -3 gpurun_out/r4y_counters.txt
rm -f gpurun_out/parity_*.log
timeout 1800 python -m pytest tests -m gpu -q --timeout 900 --timeout-method=thread 2>&1 | grep -E "passed|failed|error" | tail -3 | tee gpurun_out/r4y_gpu_tests.txt
timeout 300 python __graft_entry__.py smoke 2>&1 | grep smoke | tee gpurun_out/r4y_smoke.txt
timeout 900 python bench.py 2>gpurun_out/r4y_bench_err.txt | tail -1 > gpurun_out/r4y_bench_n1.json; python -c "
import json; d=json.load(open('gpurun_out/r4y_bench_n1.json')); print('bench', round(d['value'],4), 'frames/s', round(d['ms_per_step'],1), 'ms; igemm frac', round(d['roofline']['frac'],4), 'attention frac', round(d['roofline_attention']['frac'],4), '; configs[2]', round(d['extra']['configs[2]']['value'],3), round(d['extra']['configs[2]']['roofline']['frac'],4))" | tee gpurun_out/r4y_bench.txt
