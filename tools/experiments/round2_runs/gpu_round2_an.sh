#!/bin/bash
# round 2, run AN: appearance batch per table pass (MD_BANK_CHUNK, default 16) at configs[1], same box
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for c in 16 10 25 50 16; do
  echo "MD_BANK_CHUNK=$c $(MD_BANK_CHUNK=$c timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1)"
done > gpurun_out/r2an_bank_chunk.txt 2>&1
cat gpurun_out/r2an_bank_chunk.txt
