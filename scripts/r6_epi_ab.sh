#!/bin/bash
# round 6, step 1: the specialised conv_v4 epilogue (read-back of a round up front, branch-free passes) against the round-5 build on one box,
# and the start-time spread of the first round (conv_v4_stagger, shader cycles)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
OLD=build_ab/libuse_hip_r5.so; NEW=universal_speech_enhancement_amd/libuse_hip.so
echo "== harness + bit identity + e2e (A B A B)"; python scripts/ab_libs.py --e2e $OLD $NEW 2>&1 | tail -12
echo "== cycle stamps (new build)"; scripts/profile.sh trace build_ab/libuse_hip_trace.so 128 "L0 conv0 128->128" 2>&1 | grep "trace v4 G" | tail -44
echo "== cycle stamps +res"; scripts/profile.sh trace build_ab/libuse_hip_trace.so 128 "L0 conv1 128->128 +res" 2>&1 | grep "trace v4 G0" | tail -22
for c in "L0 conv0 128->128" "L0 conv1 128->128 +res"; do
  echo "== sustained: $c (old new old new)"; scripts/energy_ablation.sh "$c" "$OLD $NEW $OLD $NEW" 8000
  for st in 8000 16000 32000 64000; do echo "== sustained: $c new, conv_v4_stagger=$st"; EA_OPTS="--opt conv_v4_stagger=$st" scripts/energy_ablation.sh "$c" "$NEW" 8000; done
done
echo "== e2e stagger sweep"; scripts/e2e_ab.sh "conv_v4_stagger=0" "conv_v4_stagger=8000" "conv_v4_stagger=16000" "conv_v4_stagger=32000" "conv_v4_stagger=64000"
