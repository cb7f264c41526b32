#!/bin/bash
# round 5, run J: the static ring forms (65-68) against the ROUND-4 table with the tuner's cold-first-measurement bias removed (the
# base is warmed up and timed before and after the candidates)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python tools/tune_ring.py gpurun_out/igemm_tuned_r5j.inc --table tools/experiments/round5_runs/igemm_tuned_round4.inc --cfgs 65,66,67,68 --mmax 4096 > gpurun_out/r5j_tune_stream.txt 2>&1; grep -c "^M=" gpurun_out/r5j_tune_stream.txt; grep "best c6[5-8]" gpurun_out/r5j_tune_stream.txt | cut -c1-170; tail -1 gpurun_out/r5j_tune_stream.txt
