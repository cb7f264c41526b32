#!/bin/bash
# round 2, run Y: is it the LINEAR step graph that crashes rocprofv3's counter tool?  (a) separate pose, MD_OVERLAP=0 (linear graph)
# (b) merged pass + one dummy fork/join node
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R" && mkdir -p gpurun_out/prof_r2y
export TMPDIR=/tmp
cd /tmp
MD_MERGE_POSE=0 MD_OVERLAP=0 timeout 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r2y -o a_linear --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2y/a_linear.log 2>&1; echo "a_linear rc=$?"
MD_DUMMY_FORK=1 timeout 200 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r2y -o pmc_fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2y/pmc_fetch.log 2>&1; echo "b_dummy_fork fetch rc=$?"
MD_DUMMY_FORK=1 timeout 200 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r2y -o pmc_write --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2y/pmc_write.log 2>&1; echo "b_dummy_fork write rc=$?"
cd $R
python tools/summarize_profiles.py gpurun_out/prof_r2y gpurun_out/r2y_summary
rm -f gpurun_out/prof_r2y/*counter_collection.csv gpurun_out/prof_r2y/*agent_info.csv
