#!/bin/bash
# round 3, run V: same-box A/B of the tuned table (final-kernel re-tune of run U vs the committed table), two prebuilt libraries
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
{
  for lib in newtable oldtable newtable oldtable; do
    cp magicdance_amd/libmd_$lib.so.bin magicdance_amd/libmagicdance_hip.so
    echo "== $lib"
    timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
    timeout 300 python bench.py --frames-per-gpu 8 --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  done
} > gpurun_out/r3v.txt 2>&1
cat gpurun_out/r3v.txt
