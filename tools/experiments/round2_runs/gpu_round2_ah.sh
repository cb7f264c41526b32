#!/bin/bash
# round 2, run AH: final validation -- full GPU tier, smoke, default bench, 768^2 / sequence / 8-frame benches, kernel-trace stats, step breakdowns
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R" && mkdir -p gpurun_out/prof_r2ah
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r2ah_pytest.log
timeout 200 python __graft_entry__.py smoke > gpurun_out/r2ah_smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/r2ah_bench.log 2>&1
timeout 400 python bench.py --size 96 --steps 3 --warmup 1 --no-cpu-baseline --no-extra --no-roofline > gpurun_out/r2ah_bench_768.log 2>&1
timeout 400 python bench.py --sequence 16 --frames-per-gpu 8 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > gpurun_out/r2ah_bench_seq16.log 2>&1
timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r2ah_step_breakdown_1frame.txt 2>&1
timeout 300 python tools/step_breakdown.py 8 > gpurun_out/r2ah_step_breakdown_8frames.txt 2>&1
timeout 300 python tools/igemm_shape_roofline.py > gpurun_out/r2ah_igemm_shape_roofline.txt 2>&1
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r2ah -o kt --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-extra > $R/gpurun_out/prof_r2ah/kt.log 2>&1
cd $R
rm -f gpurun_out/prof_r2ah/*_kernel_trace.csv
python tools/summarize_profiles.py gpurun_out/prof_r2ah gpurun_out/r2ah_summary
tail -3 gpurun_out/r2ah_pytest.log; tail -2 gpurun_out/r2ah_smoke.log
for f in r2ah_bench r2ah_bench_768 r2ah_bench_seq16; do grep -o '"value": [0-9.]*' gpurun_out/$f.log | head -1; done
grep -o '"extra": {[^}]*}' gpurun_out/r2ah_bench.log | cut -c1-200; head -3 gpurun_out/r2ah_igemm_shape_roofline.txt
