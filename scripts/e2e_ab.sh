#!/bin/bash
# same-box end-to-end A/B of option sets: one score evaluation (configs[1] shape, two-stream schedule), 200 back-to-back evaluations each, two rounds
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for rep in 1 2; do
for o in "$@"; do
  printf "%-44s " "$o"; USE_OPTS="$o" python scripts/gpu_time_forward.py bf16 8 640 200 2>&1 | tail -1 | cut -c1-60
done; done
