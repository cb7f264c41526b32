#!/bin/bash
# round 4, run I: pipe-overlap microbenchmark; attention round-3 loop vs round-4 loop (correctness at the production shapes, unit
# tests under MD_ATTN_V=1, interleaved timing, ablations); SQ counters of both loops at 16 samples
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 120 tools/bin/pipe_bench > gpurun_out/r4i_pipe_bench.txt 2>&1; echo "pipe rc=$?"
timeout 600 python tools/attn_ab.py > gpurun_out/r4i_attn_ab.txt 2>&1; echo "ab rc=$?"
MD_ATTN_V=1 timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention and not fp8" 2>&1 | tail -3 > gpurun_out/r4i_attn_tests_v1.txt
cat gpurun_out/r4i_attn_tests_v1.txt
R=$PWD; cd /tmp
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
timeout 300 rocprofv3 --pmc $P1 --kernel-trace -d $R/gpurun_out/r4i_pmc -o p1 --output-format csv -- python $R/tools/attn_pmc4.py > $R/gpurun_out/r4i_pmc_p1.log 2>&1; echo "pmc1 rc=$?"
timeout 300 rocprofv3 --pmc $P2 --kernel-trace -d $R/gpurun_out/r4i_pmc -o p2 --output-format csv -- python $R/tools/attn_pmc4.py > $R/gpurun_out/r4i_pmc_p2.log 2>&1; echo "pmc2 rc=$?"
cd $R; ls -la gpurun_out/r4i_pmc | head; cat gpurun_out/r4i_pipe_bench.txt; cat gpurun_out/r4i_attn_ab.txt
