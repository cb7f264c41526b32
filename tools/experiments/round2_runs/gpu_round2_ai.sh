#!/bin/bash
# round 2, run AI: same-box A/B of the epilogue load schedule: MD_EPI_PIPE=0 (conditional loads, one column) vs 1 (unconditional, two columns)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for v in 0 1 0 1; do
  MD_EXTRA_FLAGS=-DMD_EPI_PIPE=$v bash magicdance_amd/csrc/build.sh > /dev/null 2>&1
  echo "== MD_EPI_PIPE=$v"
  timeout 200 python tools/igemm_epilogue_probe.py 2>&1 | grep "cfg=25\|cfg=14"
  timeout 100 python tools/launch_floor.py 2>&1 | grep "md_igemm"
  timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  timeout 300 python bench.py --frames-per-gpu 8 --steps 3 --warmup 1 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
done > gpurun_out/r2ai_epilogue_ab.txt 2>&1
cat gpurun_out/r2ai_epilogue_ab.txt
