#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_gaps; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT -o g -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $OUT/stdout.log 2>&1
echo rc=$?
python $R/scripts/trace_gaps.py $OUT/g_kernel_trace.csv
