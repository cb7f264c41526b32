#!/bin/bash
# round 5, run Y: same-box A/B of the round-4 tree (git 321591d under tools/bin/round4_tree, built from its own sources) against the FINAL
# round-5 tree (GEMM units without packed fp32), alternating bench runs, round 5 first
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
for i in 1 2 3; do
  for t in round5 round4; do
    if [ $t = round4 ]; then B=$R/tools/bin/round4_tree/bench.py; else B=$R/bench.py; fi
    (cd $(dirname $B) && timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1) | python -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d.get('extra',{}).get('configs[2]',{}); print('$t', 'frames/s', round(d['value'],4), 'ms', round(d['ms_per_step'],1), 'configs[2]', round(e.get('value',0),3))"
  done
done 2>&1 | tee gpurun_out/r5y_ab.txt
