#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
MD_FUSE_POSE=0 timeout 600 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullsize.py -m gpu -q 2>&1 | grep -E "passed|failed|^E  .*assert|FAILED" | cut -c1-300 > gpurun_out/r2g_e2e_nofuse.log
MD_FUSE_POSE=1 timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q 2>&1 | grep -E "passed|failed|^E  .*assert|FAILED" | cut -c1-300 > gpurun_out/r2g_e2e_fuse.log
MD_FUSE_POSE=1 MD_OVERLAP=0 timeout 600 python -m pytest tests/test_gpu_e2e.py -m gpu -q -k "repeated or small_b1" 2>&1 | grep -E "passed|failed|^E  .*assert|FAILED" | cut -c1-300 > gpurun_out/r2g_e2e_fuse_serial.log
echo "== nofuse"; cat gpurun_out/r2g_e2e_nofuse.log; echo "== fuse"; cat gpurun_out/r2g_e2e_fuse.log; echo "== fuse serial(overlap 0 -> old path)"; cat gpurun_out/r2g_e2e_fuse_serial.log
