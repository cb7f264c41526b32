#!/bin/bash
# round 4, run D: same-box A/B of the tuned table with / without the ring entries (two prebuilt libraries, alternating)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
cp magicdance_amd/libmagicdance_hip.so /tmp/lib_current.so
for rep in 1 2; do
  for v in r3_table r4_ring; do
    cp tools/bin/lib_$v.so magicdance_amd/libmagicdance_hip.so
    timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-extra --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', 'rep$rep', round(d['value'],4), 'frames/s', round(d['ms_per_step'],2), 'ms')" | tee -a gpurun_out/r4d_ab.txt
  done
done
cp /tmp/lib_current.so magicdance_amd/libmagicdance_hip.so
