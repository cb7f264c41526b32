#!/bin/bash
# round 3, run E: whole GPU suite (fused variant routes), kernel-trace timeline of a steady-state step, PMC attempts
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
R=$(pwd)
{
  echo "== pytest -m gpu (whole suite)"; timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8
  echo "== bench"; timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline 2>&1 | grep -o '"value": [0-9.]*' | head -1
  mkdir -p gpurun_out/r3e_prof
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace -d $R/gpurun_out/r3e_prof -o kt --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline > $R/gpurun_out/r3e_prof/bench_kt.log 2>&1; echo kt rc=$?)
  python tools/trace_gaps.py gpurun_out/r3e_prof/kt_kernel_trace.csv > gpurun_out/r3e_step_timeline.txt 2>&1; head -8 gpurun_out/r3e_step_timeline.txt
  rm -f gpurun_out/r3e_prof/kt_kernel_trace.csv
  for v in "--no-graph" ""; do
    echo "== pmc FETCH_SIZE bench $v"
    (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/r3e_prof -o pmc_fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 0 $v --no-extra --no-roofline --no-cpu-baseline > $R/gpurun_out/r3e_prof/bench_pmc_fetch.log 2>&1; echo rc=$?)
    tail -2 gpurun_out/r3e_prof/bench_pmc_fetch.log | cut -c1-200
    ls gpurun_out/r3e_prof | head
    if [ -f gpurun_out/r3e_prof/pmc_fetch_counter_collection.csv ]; then
      (cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/r3e_prof -o pmc_write --output-format csv -- python $R/bench.py --steps 1 --warmup 0 $v --no-extra --no-roofline --no-cpu-baseline > $R/gpurun_out/r3e_prof/bench_pmc_write.log 2>&1; echo rc=$?)
      python tools/summarize_profiles.py gpurun_out/r3e_prof gpurun_out/r3e_sum
      break
    fi
  done
  rm -f gpurun_out/r3e_prof/*counter_collection.csv gpurun_out/r3e_prof/*.db
} > gpurun_out/r3e.txt 2>&1
cat gpurun_out/r3e.txt
