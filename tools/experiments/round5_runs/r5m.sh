#!/bin/bash
# round 5, run M: same-box A/B of the round-4 tree (git 321591d, materialised under tools/bin/round4_tree) against this tree --
# alternating bench runs --, smoke(), then the PMC traffic passes (FETCH_SIZE / WRITE_SIZE, separate passes) of configs[1] and configs[2]
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out/prof_r5m
export TMPDIR=/tmp
R=$(pwd)
timeout 300 python __graft_entry__.py smoke 2>&1 | grep smoke | tee gpurun_out/r5m_smoke.txt
for i in 1 2; do
  for t in round4 round5; do
    if [ $t = round4 ]; then B=$R/tools/bin/round4_tree/bench.py; else B=$R/bench.py; fi
    (cd $(dirname $B) && timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1) | python -c "
import json,sys; d=json.loads(sys.stdin.read()); e=d.get('extra',{}).get('configs[2]',{}); print('$t', 'frames/s', round(d['value'],4), 'ms', round(d['ms_per_step'],1), 'configs[2]', round(e.get('value',0),3))"
  done
done 2>&1 | tee gpurun_out/r5m_ab.txt
for fpg in 1 8; do
  for c in FETCH_SIZE WRITE_SIZE; do
    tag=pmc_$( [ $c = FETCH_SIZE ] && echo fetch || echo write )
    for attempt in 1 2; do
      (cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-include-regex "igemm|ff_block" -d $R/gpurun_out/prof_r5m/f$fpg -o $tag --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-graph --no-extra --no-roofline --no-cpu-baseline --frames-per-gpu $fpg > $R/gpurun_out/prof_r5m/bench_${tag}_f$fpg.log 2>&1; echo "$tag f$fpg attempt $attempt rc=$?")
      [ -f gpurun_out/prof_r5m/f$fpg/${tag}_counter_collection.csv ] && break
    done
  done
  python tools/summarize_profiles.py gpurun_out/prof_r5m/f$fpg gpurun_out/r5m_f$fpg 2>&1 | tail -12
  rm -f gpurun_out/prof_r5m/f$fpg/*counter_collection.csv
done 2>&1 | tee gpurun_out/r5m_pmc.txt
