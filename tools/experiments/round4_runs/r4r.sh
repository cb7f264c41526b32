#!/bin/bash
# round 4, run R: the table pass on a CU-masked stream (a share of the CUs stays free for the step loop's small kernels)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r4r_cu_mask_ab.txt
F=ffffffff
for mask in "" "$F,$F,$F,$F,$F,$F,0,0" "$F,$F,$F,$F,0,0,0,0" "00ffffff,00ffffff,00ffffff,00ffffff,00ffffff,00ffffff,00ffffff,00ffffff" "0000ffff,0000ffff,0000ffff,0000ffff,0000ffff,0000ffff,0000ffff,0000ffff"; do
  for fpg in 1 8; do
    MD_TABLE_CU_MASK=$mask timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-extra --steps 4 --warmup 2 --frames-per-gpu $fpg 2>gpurun_out/r4r_err.txt | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('mask [$mask]', 'fpg$fpg', round(d['value'],4), 'frames/s', round(d['ms_per_step'],2), 'ms')
except Exception as e: print('mask [$mask] fpg$fpg failed', e)" | tee -a gpurun_out/r4r_cu_mask_ab.txt
  done
done
tail -3 gpurun_out/r4r_err.txt
