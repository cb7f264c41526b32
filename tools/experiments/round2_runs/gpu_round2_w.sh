#!/bin/bash
# round 2, run W: which workload crashes rocprofv3's counter tool (run U / V: SIGSEGV ~5 s after start)
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R" && mkdir -p gpurun_out/prof_r2w
export TMPDIR=/tmp
cd /tmp
try() { name=$1; shift; timeout 400 rocprofv3 --pmc FETCH_SIZE -d $R/gpurun_out/prof_r2w -o $name --output-format csv -- "$@" > $R/gpurun_out/prof_r2w/$name.log 2>&1; echo "$name rc=$?"; }
try a_step python $R/tools/step_breakdown.py 1
MD_MERGE_POSE=0 try b_bench_separate python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra
try c_bench_nodecode python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra --no-decode
MD_BANK_MODE=inline try d_bench_inline python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-extra
cd $R; ls -la gpurun_out/prof_r2w | head; rm -f gpurun_out/prof_r2w/*agent_info.csv
for f in gpurun_out/prof_r2w/*counter_collection.csv; do echo $f; wc -l $f; done
