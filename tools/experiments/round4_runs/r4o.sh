#!/bin/bash
# round 4, run O: first table block in growing pieces (2, 4, 10 timesteps) vs one 16-timestep pass; same box, interleaved
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/r4o_first_block_ab.txt
for rep in 1 2; do
  for sp in 0 1; do
    for fpg in 1 8; do
      MD_SPLIT_FIRST_BLOCK=$sp timeout 600 python bench.py --no-cpu-baseline --no-roofline --no-extra --steps 4 --warmup 2 --frames-per-gpu $fpg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('split_first_block $sp', 'rep$rep', 'fpg$fpg', round(d['value'],4), 'frames/s', round(d['ms_per_step'],2), 'ms')" | tee -a gpurun_out/r4o_first_block_ab.txt
    done
  done
done
timeout 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_e2e.py tests/test_gpu_rccl.py -q -x -k "baseline_config or golden or rccl or sharded or envelope" 2>&1 | tail -4
