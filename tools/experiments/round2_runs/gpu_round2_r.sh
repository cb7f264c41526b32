#!/bin/bash
# round 2, run R: LDS bank-conflict / MFMA-busy counters of the igemm k-loop (normal build), tiles 12 (128x128), 25 (128x160), 34 (128x128 32x32 frags)
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R" && mkdir -p gpurun_out/prof_r2r
export TMPDIR=/tmp PARTS_SHAPES=1 PARTS_CFGS=12,25,34
cd /tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES -d $R/gpurun_out/prof_r2r -o lds --output-format csv -- python $R/tools/igemm_parts.py > $R/gpurun_out/prof_r2r/lds.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_LDS -d $R/gpurun_out/prof_r2r -o mfma --output-format csv -- python $R/tools/igemm_parts.py > $R/gpurun_out/prof_r2r/mfma.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/prof_r2r/*counter_collection.csv")):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "igemm_kernel" not in r["Kernel_Name"]: continue
        k = (r["Kernel_Name"][:60], r["Counter_Name"])
        a = agg.setdefault(k, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
    print(f)
    for (kn, cn), (v, n) in agg.items(): print(f"  {kn:60s} {cn:32s} {v / n:14.4g} per launch ({n})")
PY
tail -3 gpurun_out/prof_r2r/lds.log
