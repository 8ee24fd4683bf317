#!/bin/bash
# bring-up: socket power / shader clock sampled while the score evaluation runs back to back (is the chip at its power limit?)
R=${GRAFT_REPO_ROOT:-$(pwd)}
(USE_OPTS="${1:-conv_v7_min_units=0}" python $R/scripts/gpu_time_forward.py bf16 8 640 ${2:-400} > $R/gpurun_out/power_run.log 2>&1) &
PID=$!
while kill -0 $PID 2>/dev/null; do
  P=$(rocm-smi --showpower 2>/dev/null | grep -oE "Power \(W\): [0-9.]+" | grep -oE "[0-9.]+$")
  C=$(rocm-smi --showclocks 2>/dev/null | grep sclk | grep -oE "\([0-9]+Mhz\)")
  echo "power $P W sclk $C"
  sleep 1
done
tail -1 $R/gpurun_out/power_run.log | cut -c1-70
