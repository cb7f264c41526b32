#!/bin/bash
# round 3, run S: the other quoted workloads on a normal box + PMC FETCH pass restricted to the igemm kernels (the counter tool
# segfaults in the first gn_small launch of the current library when it instruments every kernel)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out/prof_r3s
export TMPDIR=/tmp
R=$(pwd)
{
  timeout 300 python bench.py --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r3s_bench_n1_short.json 2>/dev/null
  timeout 300 python bench.py --size 96 --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r3s_bench_n1_768x768.json 2>/dev/null
  timeout 300 python bench.py --size 96 --fp8-attention --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r3s_bench_n1_768x768_fp8.json 2>/dev/null
  timeout 300 python bench.py --sequence 16 --frames-per-gpu 8 --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r3s_bench_n1_seq16.json 2>/dev/null
  for f in gpurun_out/r3s_bench_*.json; do echo "$f: $(grep -o '"value": [0-9.]*' $f | head -1)"; done
  (cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "igemm" -d $R/gpurun_out/prof_r3s -o pmc_fetch --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-graph --no-extra --no-roofline --no-cpu-baseline > $R/gpurun_out/prof_r3s/bench_pmc_fetch.log 2>&1; echo "fetch (igemm only) rc=$?")
  (cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "igemm" -d $R/gpurun_out/prof_r3s -o pmc_write --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-graph --no-extra --no-roofline --no-cpu-baseline > $R/gpurun_out/prof_r3s/bench_pmc_write.log 2>&1; echo "write (igemm only) rc=$?")
  python tools/summarize_profiles.py gpurun_out/prof_r3s gpurun_out/r3s_sum 2>&1 | tail -12
  rm -f gpurun_out/prof_r3s/*counter_collection.csv
  ls gpurun_out/prof_r3s
} > gpurun_out/r3s.txt 2>&1
cat gpurun_out/r3s.txt
