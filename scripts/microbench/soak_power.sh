#!/bin/bash
# Socket power / shader clock under the steady-state microbenchmark, one mode at a time (31 = MFMAs only, 23 = MFMAs + transform,
# 8 = everything but the transform, 0 = everything): scripts/microbench/soak_power.sh [seconds]
S=${1:-8}
for m in 31 23 8 0; do
  (scripts/microbench/v9_steady $m $S > /tmp/soak_$m.log 2>&1) & PID=$!
  sleep 2
  while kill -0 $PID 2>/dev/null; do
    P=$(rocm-smi --showpower 2>/dev/null | grep -oE "Power \(W\): [0-9.]+" | grep -oE "[0-9.]+$" | head -1)
    C=$(rocm-smi --showclocks 2>/dev/null | grep sclk | grep -oE "\([0-9]+Mhz\)" | head -1)
    echo "mode $m power $P W sclk $C"; sleep 1
  done
  cat /tmp/soak_$m.log
done
