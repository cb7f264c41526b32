#!/bin/bash
# round 2, run AM: same-box A/B: gn_stats 4 vs 16 loads per round trip; split-K reduce with its epilogue loads issued before / after the slab sum
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
for v in "-DMD_GN_NV=4 -DMD_REDUCE_PRELOAD=0" "-DMD_GN_NV=16 -DMD_REDUCE_PRELOAD=1" "-DMD_GN_NV=4 -DMD_REDUCE_PRELOAD=0" "-DMD_GN_NV=16 -DMD_REDUCE_PRELOAD=1"; do
  MD_EXTRA_FLAGS="$v" bash magicdance_amd/csrc/build.sh > /dev/null 2>&1
  echo "== $v"
  timeout 100 python tools/launch_floor.py 2>&1 | grep "groupnorm\|split"
  timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
done > gpurun_out/r2am_ab.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "groupnorm or splitk or two_term or second_parameter" 2>&1 | tail -2 >> gpurun_out/r2am_ab.txt
cat gpurun_out/r2am_ab.txt
