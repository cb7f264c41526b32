#!/bin/bash
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 300 python tools/igemm_epilogue_probe.py 2>&1 | grep "M=" > gpurun_out/r2aa_epilogue_probe.txt
cat gpurun_out/r2aa_epilogue_probe.txt
