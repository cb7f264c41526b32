#!/bin/bash
# round 3, run R: the profiles run Q could not finish (PMC FETCH pass: the counter tool segfaulted once), steady-state timeline and
# per-launch breakdowns of the final tree
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out/prof_r3r
export TMPDIR=/tmp
R=$(pwd)
{
  for attempt in 1 2 3; do
    (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r3r -o pmc_fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-graph --no-extra --no-roofline --no-cpu-baseline > $R/gpurun_out/prof_r3r/bench_pmc_fetch.log 2>&1; echo "fetch attempt $attempt rc=$?")
    [ -f gpurun_out/prof_r3r/pmc_fetch_counter_collection.csv ] && break
  done
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/prof_r3r -o kt --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline > $R/gpurun_out/prof_r3r/bench_kt.log 2>&1; echo kt rc=$?)
  python tools/trace_gaps.py gpurun_out/prof_r3r/kt_kernel_trace.csv > gpurun_out/r3r_step_timeline.txt 2>&1; head -4 gpurun_out/r3r_step_timeline.txt
  rm -f gpurun_out/prof_r3r/kt_kernel_trace.csv
  timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r3r_step_breakdown_1frame.txt 2>&1
  timeout 300 python tools/step_breakdown.py 8 > gpurun_out/r3r_step_breakdown_8frames.txt 2>&1
  ls -la gpurun_out/prof_r3r
} > gpurun_out/r3r.txt 2>&1
cat gpurun_out/r3r.txt
