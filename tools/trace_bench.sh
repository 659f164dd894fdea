#!/bin/bash
# kernel trace of a short headline run (no secondary lines), then the timeline of its last step
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_b
timeout -k 10 400 rocprofv3 --output-format csv --kernel-trace -d $R/gpurun_out/trace_b -o t -- python3 $R/bench.py --steps 2 --warmup 1 --no-secondary --no-cpu-baseline > $R/gpurun_out/trace_b.log 2>&1 || exit 1
tail -1 $R/gpurun_out/trace_b.log | cut -c1-200
python3 $R/tools/timeline.py $(find $R/gpurun_out/trace_b -name "*kernel_trace.csv" | head -1)
