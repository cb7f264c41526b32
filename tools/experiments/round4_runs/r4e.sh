#!/bin/bash
# round 4, run E: per-launch step breakdown (un-captured, event-timed) with the round-3 table and with the ring entries, same box
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
cp magicdance_amd/libmagicdance_hip.so /tmp/lib_current.so
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -v "^$" | head -20 > gpurun_out/r4e_smi.txt
for v in r3_table r4_ring; do
  cp tools/bin/lib_$v.so magicdance_amd/libmagicdance_hip.so
  timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r4e_breakdown_$v.txt 2>/dev/null
  head -3 gpurun_out/r4e_breakdown_$v.txt
done
cp /tmp/lib_current.so magicdance_amd/libmagicdance_hip.so
