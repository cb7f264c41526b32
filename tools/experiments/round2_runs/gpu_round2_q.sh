#!/bin/bash
# round 2, run Q: component timing of the igemm k-loop (debug build on the box): masks 0 full, 1 no MFMA (LDS reads only),
# 2 no LDS reads + MFMA (loads only), 4 no k-loop loads (compute only), 5 LDS reads only without loads
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
MD_EXTRA_FLAGS=-DMD_IGEMM_DEBUG bash magicdance_amd/csrc/build.sh > gpurun_out/r2q_build.log 2>&1
for m in 0 4; do MD_IGEMM_DEBUG=$m timeout 200 python tools/igemm_parts.py 2>&1 | grep dbg; done > gpurun_out/r2q_igemm_parts.txt
cat gpurun_out/r2q_igemm_parts.txt
