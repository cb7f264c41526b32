#!/bin/bash
# round 5, run K: static ring at the 64 x 64 level too (A blocks up to 288 rows): parity (the c3_64 case), tuner on 4096 < M <= 16384,
# then the step / bench A-B of the new table (MD_IGEMM_TUNED unchanged; old table = tools/experiments/round5_runs/igemm_tuned_round4.inc)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_igemm_ring.py -q -x --timeout 600 -k "(65 or 66) and (c3_64 or config_table or c3_32)" 2>&1 | tail -4 | tee gpurun_out/r5k_tests.txt
timeout 1500 python tools/tune_ring.py gpurun_out/igemm_tuned_r5k.inc --cfgs 65,66 --mmin 4097 --mmax 16384 > gpurun_out/r5k_tune_stream.txt 2>&1; grep -c "^M=" gpurun_out/r5k_tune_stream.txt; grep "ks=3" gpurun_out/r5k_tune_stream.txt | cut -c1-200; tail -1 gpurun_out/r5k_tune_stream.txt
timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r5k_step_1f.txt 2>&1; head -3 gpurun_out/r5k_step_1f.txt | tail -2
for i in 1 2; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-extra 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new table frames/s', round(d['value'],4), 'ms', round(d['ms_per_step'],1))"
done 2>&1 | tee gpurun_out/r5k_bench.txt
