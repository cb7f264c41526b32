#!/bin/bash
# round 5, run X: PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate passes) of configs[1] and configs[2] on the FINAL library
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out/prof_r5x
export TMPDIR=/tmp
R=$(pwd)
for fpg in 1 8; do
  for c in FETCH_SIZE WRITE_SIZE; do
    tag=pmc_$( [ $c = FETCH_SIZE ] && echo fetch || echo write )
    (cd /tmp && timeout 420 rocprofv3 --pmc $c --kernel-include-regex "igemm|ff_block" -d $R/gpurun_out/prof_r5x/f$fpg -o $tag --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-graph --no-extra --no-roofline --no-cpu-baseline --frames-per-gpu $fpg > $R/gpurun_out/prof_r5x/bench_${tag}_f$fpg.log 2>&1; echo "$tag f$fpg rc=$?")
  done
  python tools/summarize_profiles.py gpurun_out/prof_r5x/f$fpg gpurun_out/r5x_f$fpg 2>&1 | tail -12
  rm -f gpurun_out/prof_r5x/f$fpg/*counter_collection.csv
done 2>&1 | tee gpurun_out/r5x_pmc.txt
