#!/bin/bash
# timing-only upper bounds: one score evaluation (configs[1] shape, graph + two streams) with classes of launches dropped
L=scripts/ab/libuse_hip_skip.so
run() { echo "$1: $(env $2 USE_HIP_LIB=$L python scripts/gpu_time_forward.py bf16 8 640 5 2>&1 | tail -1 | cut -c1-45)"; }
for r in 1 2; do
run "full            " "X=1"
run "no L6 (<=80)    " "USE_HIP_SKIP=0:80"
run "no L5-6 (<=320) " "USE_HIP_SKIP=0:320"
run "no L4-6 (<=1280)" "USE_HIP_SKIP=0:1280"
run "no L3-6 (<=5120)" "USE_HIP_SKIP=0:5120"
run "no L2-6 (<=20480)" "USE_HIP_SKIP=0:20480"
run "no L2 only      " "USE_HIP_SKIP=5121:20480"
run "no FIR          " "USE_HIP_SKIP_FIR=1"
done
