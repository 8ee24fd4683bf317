#!/bin/bash
# cycle stamps of conv_v4 and conv_v5 workgroups: a lone launch and inside the evaluation (trace builds: scripts/build_variant.sh trace / VARIANT_FILE=use_conv_v5 ... trace5)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for c in "L0 conv0 128->128" "L0 conv1 128->128 +res"; do for v in 4 5; do
  L=build_ab/libuse_hip_trace.so; [ $v = 5 ] && L=build_ab/libuse_hip_trace5.so
  echo "== v$v $c (lone launch)"
  USE_HIP_LIB=$R/$L USE_HIP_TRACE=128 python scripts/gpu_conv_trace.py $v "$c" 2>&1 | grep "trace v4 G" | awk '{printf "%s %s:%s ", ($5=="1"?"\n"$3:""), $5, $7} END{print ""}'
done; done
echo "== conv_v5 inside the evaluation"; TRACE_LIB=build_ab/libuse_hip_trace5.so scripts/r6_trace_in_situ.sh 2>&1 | grep -A1 "#64"
echo "== conv_v4 inside the evaluation"; USE_OPTS=conv_v5=0 TRACE_LIB=build_ab/libuse_hip_trace.so scripts/r6_trace_in_situ.sh 2>&1 | grep -A1 "#64"
