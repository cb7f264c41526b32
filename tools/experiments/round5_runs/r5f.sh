#!/bin/bash
# round 5, run F: SQ counters of md_ff_block in isolation (MFMA busy, wave-cycle split, VALU / SALU / LDS instruction counts)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out/ffpmc
export TMPDIR=/tmp
R=$(pwd); D=$R/gpurun_out/ffpmc
{
  (cd /tmp && timeout 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-include-regex ff_block -d "$D" -o sq --output-format csv -- python $R/tools/ffblock_pmc.py > "$D/sq.log" 2>&1; echo sq rc=$?)
  (cd /tmp && timeout 240 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --kernel-include-regex ff_block -d "$D" -o in --output-format csv -- python $R/tools/ffblock_pmc.py > "$D/in.log" 2>&1; echo in rc=$?)
  (cd /tmp && timeout 240 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL --kernel-include-regex ff_block -d "$D" -o m3 --output-format csv -- python $R/tools/ffblock_pmc.py > "$D/m3.log" 2>&1; echo m3 rc=$?)
  python - "$D" <<'PY'
import csv, sys, collections, os
d = sys.argv[1]
for tag in ("sq", "in", "m3"):
    p = os.path.join(d, f"{tag}_counter_collection.csv")
    if not os.path.exists(p):
        print(tag, "missing"); continue
    rows = list(csv.DictReader(open(p)))
    byk = collections.OrderedDict()
    for r in rows:
        key = (r["Dispatch_Id"], r["Kernel_Name"].replace("(anonymous namespace)::", "")[:60], r["Grid_Size"])
        byk.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for k, v in byk.items():
        print(tag, k[0], k[1], "grid", k[2], " ".join(f"{a}={b:.6g}" for a, b in v.items()))
PY
  rm -f "$D"/*counter_collection.csv
} > gpurun_out/r5f_ffblock_counters.txt 2>&1
tail -40 gpurun_out/r5f_ffblock_counters.txt
