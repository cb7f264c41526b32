#!/bin/bash
# round 3, run P: full validation of the tree (all GPU tests, smoke, default bench + quoted workloads, profiles)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_fullsize.log
{
  echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
} > gpurun_out/r3p.txt 2>&1
timeout 900 python bench.py > gpurun_out/r3p_bench_n1.json 2> gpurun_out/r3p_bench_n1.err
timeout 300 python bench.py --size 96 --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r3p_bench_n1_768x768.json 2>/dev/null
timeout 300 python bench.py --size 96 --fp8-attention --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r3p_bench_n1_768x768_fp8.json 2>/dev/null
timeout 300 python bench.py --sequence 16 --frames-per-gpu 8 --steps 2 --warmup 1 --no-extra --no-roofline --no-cpu-baseline > gpurun_out/r3p_bench_n1_seq16.json 2>/dev/null
for f in gpurun_out/r3p_bench_*.json; do echo "$f: $(grep -o '"value": [0-9.]*' $f | head -2 | tr '\n' ' ')" >> gpurun_out/r3p.txt; done
grep -o '"frac": [0-9.]*' gpurun_out/r3p_bench_n1.json >> gpurun_out/r3p.txt
bash tools/run_profiles.sh prof_r3p >> gpurun_out/r3p.txt 2>&1
timeout 300 python tools/step_breakdown.py 1 > gpurun_out/r3p_step_breakdown_1frame.txt 2>&1
timeout 300 python tools/step_breakdown.py 8 > gpurun_out/r3p_step_breakdown_8frames.txt 2>&1
cat gpurun_out/r3p.txt
