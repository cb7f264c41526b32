#!/bin/bash
# round 4, run N: stream priority of the step loop over the table pass (one-frame and 8-frame batches), same box, interleaved
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r4n_prio_ab.txt
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')" 2>/dev/null | tee -a gpurun_out/r4n_prio_ab.txt
for rep in 1 2; do
  for pr in 0 -1; do
    for fpg in 1 8; do
      MD_STEP_PRIORITY=$pr timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-extra --steps 4 --warmup 2 --frames-per-gpu $fpg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('prio $pr', 'rep$rep', 'fpg$fpg', round(d['value'],4), 'frames/s', round(d['ms_per_step'],2), 'ms')" | tee -a gpurun_out/r4n_prio_ab.txt
    done
  done
done
