#!/bin/bash
# round 2, run V: PMC fetch / write passes (retry; run U's rocprofv3 counter tool crashed at start-up)
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R" && mkdir -p gpurun_out/prof_r2v
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r2v -o pmc_fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2v/pmc_fetch.log 2>&1
echo "fetch rc=$?"
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $R/gpurun_out/prof_r2v -o pmc_write --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2v/pmc_write.log 2>&1
echo "write rc=$?"
cd $R
python tools/summarize_profiles.py gpurun_out/prof_r2v gpurun_out/r2v_summary
rm -f gpurun_out/prof_r2v/*counter_collection.csv
ls gpurun_out/
