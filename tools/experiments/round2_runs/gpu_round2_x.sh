#!/bin/bash
# round 2, run X: PMC WRITE_SIZE / FETCH_SIZE with the pose ControlNet on its own stream (MD_MERGE_POSE=0): rocprofv3's counter tool
# crashes (SIGSEGV in the tool, runs U / V) or hangs (run W) on the default merged-pass step graph in table mode; kernel-trace is fine
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R" && mkdir -p gpurun_out/prof_r2x
export TMPDIR=/tmp MD_MERGE_POSE=0
cd /tmp
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r2x -o pmc_fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2x/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r2x -o pmc_write --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2x/pmc_write.log 2>&1; echo "write rc=$?"
cd $R
python tools/summarize_profiles.py gpurun_out/prof_r2x gpurun_out/r2x_summary
rm -f gpurun_out/prof_r2x/*counter_collection.csv gpurun_out/prof_r2x/*agent_info.csv
