#!/bin/bash
# round 5, run I: the static 1x1 / linear ring form (configs 67 / 68): parity tests, tuner on M <= 1024 (1x1) and on the 3x3 convs of
# 1024 < M <= 4096 (configs 65 / 66)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_igemm_ring.py -q -x --timeout 600 -k "67 or 68 or config_table" 2>&1 | tail -8 | tee gpurun_out/r5i_stream1_tests.txt
timeout 900 python tools/tune_ring.py gpurun_out/igemm_tuned_r5i.inc --cfgs 65,66,67,68 --mmax 4096 > gpurun_out/r5i_tune_stream.txt 2>&1; grep -c "^M=" gpurun_out/r5i_tune_stream.txt; grep "best c6[5-8]" gpurun_out/r5i_tune_stream.txt | cut -c1-170; tail -1 gpurun_out/r5i_tune_stream.txt
