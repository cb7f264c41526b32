#!/bin/bash
# GPU box: PMC counters of the attention kernels (d=40, 64^2, cond + uncond batched) -- separate passes, --pmc only
cd "$(dirname "$0")/.." && mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/pmc/counters_list.txt 2>&1
run() {  # name, env, counters...
  name=$1; shift; envs=$1; shift
  env $envs ATTN_BENCH_ONLY=0 ATTN_BENCH_REPS=2 timeout 300 rocprofv3 --pmc "$@" -d $GRAFT_REPO_ROOT/gpurun_out/pmc/$name -o $name --output-format csv -- python $GRAFT_REPO_ROOT/tools/attn_bench.py > $GRAFT_REPO_ROOT/gpurun_out/pmc/$name.log 2>&1
}
for v in v2 v3; do
  if [ $v = v3 ]; then E="MD_ATTN_V=3 MD_ATTN_P=1"; else E="MD_ATTN_V=2"; fi
  run ${v}_a "$E" SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  run ${v}_b "$E" SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU
  run ${v}_c "$E" SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD
  run ${v}_d "$E" GRBM_GUI_ACTIVE GRBM_COUNT TCC_HIT_sum TCC_MISS_sum
done
cd $GRAFT_REPO_ROOT/gpurun_out/pmc; ls -R | head -40; for f in *.log; do echo == $f; tail -3 $f; done
