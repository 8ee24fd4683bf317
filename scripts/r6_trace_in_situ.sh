#!/bin/bash
# cycle stamps of one conv_v4 workgroup INSIDE the three-stream evaluation (the other sub-batches' kernels on the chip), next to the same
# launch alone on the chip: how long prologue / chunks / epilogue last when the CUs are not in lock step
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for res in 0 1; do for skip in 0 30 47 64; do
  echo "== 128->128 at 512x640, res=$res, matching launch #$skip, workgroup 900"
  USE_HIP_LIB=$R/${TRACE_LIB:-build_ab/libuse_hip_trace.so} USE_HIP_TRACE=128 USE_HIP_TRACE_RES=$res USE_HIP_TRACE_SKIP=$skip USE_HIP_TRACE_WG=900 python scripts/gpu_time_forward.py bf16 8 640 12 2>&1 | grep "trace v4 G0" | awk '{printf "%s:%s ", $5, $7} END{print ""}'
done; done
