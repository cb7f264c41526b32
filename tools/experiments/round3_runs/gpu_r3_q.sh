#!/bin/bash
# round 3, run Q: repeat of run P on another box (P's box ran every kernel 15-100 % slower than the boxes of runs J..O): all GPU
# tests with the summary line kept, default bench, profiles
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
rm -f gpurun_out/parity_fullsize.log
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|power" | head -6 > gpurun_out/r3q.txt
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl\|NCCL\|^$" | tail -6 >> gpurun_out/r3q.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 >> gpurun_out/r3q.txt
timeout 900 python bench.py > gpurun_out/r3q_bench_n1.json 2> gpurun_out/r3q_bench_n1.err
grep -o '"value": [0-9.]*' gpurun_out/r3q_bench_n1.json | head -2 >> gpurun_out/r3q.txt
grep -o '"frac": [0-9.]*' gpurun_out/r3q_bench_n1.json >> gpurun_out/r3q.txt
(cd /tmp; timeout 200 rocprofv3 --pmc FETCH_SIZE -d /tmp/pmcprobe -o p --output-format csv -- python $OLDPWD/tools/pmc_probe.py 27 1 3 > $OLDPWD/gpurun_out/r3q_pmc_probe.txt 2>&1; echo "pmc_probe rc=$?" >> $OLDPWD/gpurun_out/r3q.txt)
bash tools/run_profiles.sh prof_r3q >> gpurun_out/r3q.txt 2>&1
cat gpurun_out/r3q.txt
