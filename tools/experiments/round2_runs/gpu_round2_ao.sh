#!/bin/bash
# round 2, run AO: last validation of the final tree: full GPU tier + smoke
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 > gpurun_out/r2ao_pytest.log
timeout 200 python __graft_entry__.py smoke > gpurun_out/r2ao_smoke.log 2>&1
tail -1 gpurun_out/r2ao_pytest.log; tail -2 gpurun_out/r2ao_smoke.log
