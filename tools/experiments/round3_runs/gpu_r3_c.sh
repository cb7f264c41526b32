#!/bin/bash
# round 3, run C: whole GPU suite on the finalize + apply GroupNorm / tuned k-group table build, A/B of the GroupNorm fusion,
# kernel-trace + PMC (FETCH / WRITE on the un-captured timed mix) profiles.
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
{
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
  for env in "MD_GN_FUSE=1" "MD_GN_FUSE=0" "MD_GN_FUSE=1" "MD_GN_FUSE=0"; do
    echo "== $env"
    env $env timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
    env $env timeout 300 python bench.py --frames-per-gpu 8 --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  done
} > gpurun_out/r3c.txt 2>&1
timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r3c_step_breakdown_1frame.txt 2>&1
timeout 300 python tools/step_breakdown.py 8 > gpurun_out/r3c_step_breakdown_8frames.txt 2>&1
bash tools/run_profiles.sh r3c_prof > gpurun_out/r3c_prof.log 2>&1
python tools/summarize_profiles.py gpurun_out/r3c_prof gpurun_out/r3c_sum >> gpurun_out/r3c_prof.log 2>&1
rm -f gpurun_out/r3c_prof/*counter_collection.csv gpurun_out/r3c_prof/*.db
tail -25 gpurun_out/r3c_prof.log >> gpurun_out/r3c.txt
cat gpurun_out/r3c.txt
