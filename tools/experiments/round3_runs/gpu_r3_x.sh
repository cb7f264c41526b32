#!/bin/bash
# round 3, run X: rows per reference-KV table pass (appearance batch) at one frame: 16 (-> passes of 16/16/16/2), 17 (17/17/16), 25, 50
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
{
  for c in 16 17 25 50 16 25; do
    echo "== MD_BANK_CHUNK=$c"
    MD_BANK_CHUNK=$c timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  done
  echo "== 8 frames"
  for c in 16 25; do
    echo "== MD_BANK_CHUNK=$c"; MD_BANK_CHUNK=$c timeout 300 python bench.py --frames-per-gpu 8 --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  done
} > gpurun_out/r3x.txt 2>&1
cat gpurun_out/r3x.txt
