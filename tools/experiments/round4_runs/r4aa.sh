#!/bin/bash
# round 4, run AA: the round's tracked profiles again on the FINAL tree (after the shared LayerNorm statistics): step breakdowns (1 / 8 frames),
# rocprofv3 kernel stats + PMC traffic passes (tools/run_profiles.sh)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r4aa_step_breakdown_1frame.txt 2>&1; head -3 gpurun_out/r4aa_step_breakdown_1frame.txt | tail -2
timeout 300 python tools/step_breakdown.py 8 > gpurun_out/r4aa_step_breakdown_8frames.txt 2>&1; head -3 gpurun_out/r4aa_step_breakdown_8frames.txt | tail -2
bash tools/run_profiles.sh r4aa_prof 2>&1 | tail -8
python tools/summarize_profiles.py gpurun_out/r4aa_prof gpurun_out/r4aa_summary 2>&1 | tail -3
