#!/bin/bash
# same-box end-to-end A/B of option sets: one score evaluation (configs[1] shape unless E2E_B / E2E_PREC say otherwise, multi-stream schedule),
# 200 back-to-back evaluations each, two rounds.   [E2E_B=16 E2E_N=100] scripts/e2e_ab.sh "opt=v,opt=v" "..." ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for rep in 1 2; do
for o in "$@"; do
  printf "%-44s " "$o"; USE_OPTS="$o" python scripts/gpu_time_forward.py ${E2E_PREC:-bf16} ${E2E_B:-8} 640 ${E2E_N:-200} 2>&1 | tail -1 | cut -c1-60
done; done
