#!/bin/bash
# same-box A/B of library builds: the evaluation (bf16, configs[1] shape, 200 evaluations each, two rounds) and sustained launches of three convolutions
#   scripts/r6_lib_ab.sh libA.so libB.so [...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for rep in 1 2; do for l in "$@"; do printf "%-46s " "$l"; USE_HIP_LIB=$R/$l python scripts/gpu_time_forward.py bf16 8 640 200 2>&1 | tail -1 | cut -c1-60; done; done
for c in "L0 conv0 128->128" "L0 conv1 128->128 +res" "L1 conv0 cat384->128"; do echo "== sustained: $c"; VARIANT=0 scripts/energy_ablation.sh "$c" "$* $*" 8000; done
