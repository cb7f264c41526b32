#!/bin/bash
# round 5, run W: SQ counters of the low-resolution self + bank attention launches (d = 80 at 32x32, d = 160 at 16x16; review item 6)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out/attnpmc
export TMPDIR=/tmp
R=$(pwd); D=$R/gpurun_out/attnpmc
{
  (cd /tmp && timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-include-regex attn_kernel -d "$D" -o sq --output-format csv -- python $R/tools/attn_pmc_lowres.py > "$D/sq.log" 2>&1; echo sq rc=$?)
  (cd /tmp && timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-include-regex attn_kernel -d "$D" -o in --output-format csv -- python $R/tools/attn_pmc_lowres.py > "$D/in.log" 2>&1; echo in rc=$?)
  (cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --kernel-include-regex attn_kernel -d "$D" -o kt --output-format csv -- python $R/tools/attn_pmc_lowres.py > "$D/kt.log" 2>&1; echo kt rc=$?)
  python - "$D" <<'PY'
import csv, sys, collections, os
d = sys.argv[1]
for tag in ("sq", "in"):
    p = os.path.join(d, f"{tag}_counter_collection.csv")
    if not os.path.exists(p):
        print(tag, "missing"); continue
    rows = list(csv.DictReader(open(p)))
    byk = collections.OrderedDict()
    for r in rows:
        key = (int(r["Dispatch_Id"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")[:40], r["Grid_Size"])
        byk.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for k, v in sorted(byk.items()):
        print(tag, k[0], k[1], "grid", k[2], " ".join(f"{a}={b:.6g}" for a, b in v.items()))
p = os.path.join(d, "kt_kernel_trace.csv")
if os.path.exists(p):
    rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Start_Timestamp"]))
    for i, r in enumerate(rows):
        print("kt", i, r["Kernel_Name"].replace("(anonymous namespace)::", "")[:40], "grid", r.get("Grid_Size_X", r.get("Grid_Size", "?")), "us", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
PY
  rm -f "$D"/*counter_collection.csv "$D"/kt_kernel_trace.csv
} > gpurun_out/r5w_attention_lowres_counters.txt 2>&1
tail -5 gpurun_out/r5w_attention_lowres_counters.txt
