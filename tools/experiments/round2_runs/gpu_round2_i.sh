#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 tools/bin/mfma_bench > gpurun_out/r2i_mfma_issue_rate.txt 2>&1
timeout 600 python tools/igemm_shape_roofline.py > gpurun_out/r2i_igemm_shape_roofline.txt 2>&1
cat gpurun_out/r2i_mfma_issue_rate.txt; head -30 gpurun_out/r2i_igemm_shape_roofline.txt
