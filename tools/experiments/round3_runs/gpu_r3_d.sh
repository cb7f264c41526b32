#!/bin/bash
# round 3, run D: which md_igemm variants does `rocprofv3 --pmc` survive?  + weight-prefetch A/B + remaining GPU tests
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
{
  for cg in "27 1" "24 1" "24 2" "33 1" "15 4" "12 1" "25 1" "7 1"; do
    echo "== pmc probe cfg/kg $cg"
    (cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE -d /tmp/pp -o p --output-format csv -- python $OLDPWD/tools/pmc_probe.py $cg 2>&1 | grep -E "^ok|SIGSEGV|Segmentation|dumped" | head -3)
  done
  for env in "MD_PREFETCH=1" "MD_PREFETCH=0" "MD_PREFETCH=1" "MD_PREFETCH=0"; do
    echo "== $env"
    env $env timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
    env $env timeout 300 python bench.py --frames-per-gpu 8 --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  done
  echo "== pytest -m gpu (whole suite)"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8
} > gpurun_out/r3d.txt 2>&1
cat gpurun_out/r3d.txt
