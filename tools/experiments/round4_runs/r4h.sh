#!/bin/bash
# round 4, run H: whole GPU test tier with the final ring table, then same-box A/B of the two tables (1 frame and 8 frames)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 --timeout-method=thread 2>&1 | tail -5 > gpurun_out/r4h_gpu_tests.txt
cat gpurun_out/r4h_gpu_tests.txt
cp magicdance_amd/libmagicdance_hip.so /tmp/lib_current.so
rm -f gpurun_out/r4h_ab.txt
for rep in 1 2; do
  for v in r3_table r4_ring; do
    cp tools/bin/lib_$v.so magicdance_amd/libmagicdance_hip.so
    for fpg in 1 8; do
      timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-extra --steps 4 --warmup 2 --frames-per-gpu $fpg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('$v', 'rep$rep', 'fpg$fpg', round(d['value'],4), 'frames/s', round(d['ms_per_step'],2), 'ms')" | tee -a gpurun_out/r4h_ab.txt
    done
  done
done
cp /tmp/lib_current.so magicdance_amd/libmagicdance_hip.so
