#!/bin/bash
# round 4, run K: the round-4 attention loop on the other production shapes
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out
timeout 600 python tools/attn_ab3.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4k_attn_shapes.txt; cat gpurun_out/r4k_attn_shapes.txt
