#!/bin/bash
# round 6: conv_v5 (16x16x32 MFMAs) against conv_v4 (32x32x16), one box: sustained launches (joules), the evaluation (option conv_v5 = 0 / 1), bf16 and fp16
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
LIB=universal_speech_enhancement_amd/libuse_hip.so
for c in "L0 conv0 128->128" "L0 conv1 128->128 +res" "L1 conv0 cat384->128" "L2 conv0 256->256"; do
  for v in 4 5 4 5; do echo -n "v$v $c: "; VARIANT=$v scripts/energy_ablation.sh "$c" "$LIB" 8000; done
done
echo "== evaluation, bf16"; scripts/e2e_ab.sh conv_v5=0 conv_v5=1
echo "== evaluation, fp16"; E2E_PREC=fp16 scripts/e2e_ab.sh conv_v5=0 conv_v5=1
