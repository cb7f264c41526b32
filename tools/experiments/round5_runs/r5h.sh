#!/bin/bash
# round 5, run H: the static ring form of md_igemm (configs 65 / 66, igemm_stream.hip): parity tests, then the tuner against the
# committed table on the M <= 1024 3x3 layers
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_igemm_ring.py -q -x --timeout 600 -k "65 or 66 or config_table" 2>&1 | tail -8 | tee gpurun_out/r5h_stream_tests.txt
timeout 900 python tools/tune_ring.py gpurun_out/igemm_tuned_r5h.inc --cfgs 65,66 --mmax 1024 > gpurun_out/r5h_tune_stream.txt 2>&1; tail -45 gpurun_out/r5h_tune_stream.txt
