#!/bin/bash
# Round profiles of the headline bench (GPU box only):  bash tools/run_profiles.sh <out dir under gpurun_out>
# 1. rocprofv3 --kernel-trace --stats of `bench.py --steps 2 --warmup 1` (configs[1])
# 2./3. separate PMC passes (FETCH_SIZE, WRITE_SIZE) on ONE un-captured batch of the same workload, never combined with traces
# Summaries for profiles/ are produced afterwards by tools/summarize_profiles.py <dir> profiles/roundN
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
D=$R/gpurun_out/${1:-prof}
mkdir -p "$D"
cd /tmp && export TMPDIR=/tmp
cd "$R"
B="--no-cpu-baseline --no-roofline"
timeout 400 rocprofv3 --kernel-trace --stats -d "$D" -o kt --output-format csv -- python bench.py --steps 2 --warmup 1 --no-extra $B > "$D/bench_kt.log" 2>&1; echo kt rc=$?
# the PMC passes run the benchmarked batch itself (table pass + 50 steps + decode, the default merged pass) with every launch
# un-captured (--no-graph): same launches, same arguments; the counter tool does not survive replays of the linear step graph
# (round 3: counters are collected for the igemm kernels only -- --kernel-include-regex; instrumenting every kernel, the counter tool
#  of this image segfaults in the first gn_small launch of the current library, runs P / Q / R)
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex igemm -d "$D" -o pmc_fetch --output-format csv -- python bench.py --steps 1 --warmup 0 --no-graph --no-extra $B > "$D/bench_pmc_fetch.log" 2>&1; echo fetch rc=$?
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex igemm -d "$D" -o pmc_write --output-format csv -- python bench.py --steps 1 --warmup 0 --no-graph --no-extra $B > "$D/bench_pmc_write.log" 2>&1; echo write rc=$?
rm -f "$D"/kt_kernel_trace.csv   # per-launch trace: too large to merge back; the stats CSV carries what profiles/ needs
ls -la "$D"
