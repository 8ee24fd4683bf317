#!/bin/bash
# same-box A/B of two library builds on one score evaluation at configs[1]: ab_forward.sh <libA.so> <libB.so> [rounds]
A=$1; B=$2; R=${3:-3}
for r in $(seq $R); do
  echo "A $(USE_HIP_LIB=$A python scripts/gpu_time_forward.py bf16 8 640 5 2>&1 | tail -1 | cut -c1-60)"
  echo "B $(USE_HIP_LIB=$B python scripts/gpu_time_forward.py bf16 8 640 5 2>&1 | tail -1 | cut -c1-60)"
done
