#!/bin/bash
# round 6: conv_v4 epilogue variants against earlier builds on one box: harness + bit identity + end to end, cycle stamps, sustained joules
#   scripts/r6_epi_ab.sh [old.so ...]   (default: the round-5 build and the first round-6 epilogue)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
OLDS=${@:-build_ab/libuse_hip_r5.so build_ab/libuse_hip_r6a.so}; NEW=universal_speech_enhancement_amd/libuse_hip.so
echo "== harness + bit identity + e2e"; python scripts/ab_libs.py --e2e $OLDS $NEW 2>&1 | tail -14
echo "== cycle stamps (new build), lone launch"; scripts/profile.sh trace build_ab/libuse_hip_trace.so 128 "L0 conv0 128->128" 2>&1 | grep "trace v4 G" | awk '{printf "%s %s:%s ", ($5=="1"?"\n"$3:""), $5, $7} END{print ""}'
echo "== cycle stamps +res"; scripts/profile.sh trace build_ab/libuse_hip_trace.so 128 "L0 conv1 128->128 +res" 2>&1 | grep "trace v4 G" | awk '{printf "%s %s:%s ", ($5=="1"?"\n"$3:""), $5, $7} END{print ""}'
for c in "L0 conv0 128->128" "L0 conv1 128->128 +res" "L1 conv0 cat384->128"; do
  echo "== sustained: $c"; scripts/energy_ablation.sh "$c" "$OLDS $NEW $OLDS $NEW" 8000
done
