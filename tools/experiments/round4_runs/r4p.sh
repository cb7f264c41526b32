#!/bin/bash
# round 4, run P: kernel-trace timeline of whole batches (where does a one-frame batch spend the time that is not step kernels?)
cd "$(dirname "$0")/../../.." && mkdir -p gpurun_out/r4p
export TMPDIR=/tmp
R=$PWD; cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/r4p -o kt --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-extra --no-cpu-baseline --no-roofline > $R/gpurun_out/r4p/bench.log 2>&1; echo rc=$?
cd $R
tail -1 gpurun_out/r4p/bench.log | cut -c1-200
python tools/batch_timeline.py gpurun_out/r4p/kt_kernel_trace.csv | tee gpurun_out/r4p_batch_timeline_1frame.txt
python tools/trace_gaps.py gpurun_out/r4p/kt_kernel_trace.csv > gpurun_out/r4p_step_timeline_1frame.txt; head -4 gpurun_out/r4p_step_timeline_1frame.txt
rm -f gpurun_out/r4p/kt_kernel_trace.csv
