#!/bin/bash
# round 3, run W: SQ / L2 counters of isolated md_igemm shapes on the final kernels (counter passes restricted to the igemm kernels)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out/igpmc_r3w
export TMPDIR=/tmp
R=$(pwd); D=$R/gpurun_out/igpmc_r3w
{
  (cd /tmp && timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-include-regex igemm -d "$D" -o sq --output-format csv -- python $R/tools/igemm_pmc.py > "$D/sq.log" 2>&1; echo sq rc=$?)
  (cd /tmp && timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum --kernel-include-regex igemm -d "$D" -o l2 --output-format csv -- python $R/tools/igemm_pmc.py > "$D/l2.log" 2>&1; echo l2 rc=$?)
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
        if "igemm_kernel" not in r["Kernel_Name"]:
            continue
        key = (r["Dispatch_Id"], r["Kernel_Name"].replace("(anonymous namespace)::", "")[:70], r["Grid_Size"])
        byk.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for k, v in byk.items():
        print(tag, k[1], "grid", k[2], " ".join(f"{a}={b:.4g}" for a, b in v.items()))
PY
  rm -f "$D"/*counter_collection.csv
} > gpurun_out/r3w.txt 2>&1
cat gpurun_out/r3w.txt
