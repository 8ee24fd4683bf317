#!/bin/bash
# Build an A/B variant of the library with extra -D flags on ONE kernel file (VARIANT_FILE, default use_conv_v4):
#   [VARIANT_FILE=use_conv_v2] scripts/build_variant.sh NAME [-DFLAG ...]
# -> build_ab/libuse_hip_NAME.so  (git-ignored; for scripts/ab_libs.py / scripts/profile.sh ab|trace)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/universal_speech_enhancement_amd/csrc; N=$1; shift; F=${VARIANT_FILE:-use_conv_v4}
mkdir -p $R/build_ab
make -s -C $C -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $C/$F.hip -o $R/build_ab/variant_$N.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "conv_v4_kernelIDF16bS0_Li32ELb1E\|error" | grep -i "error\|VGPRs:\|Spill\|LDS Size" | head -8 || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_ab/libuse_hip_$N.so $(for o in use_kernels use_conv_v2 use_conv_v4 use_conv_v5 use_attn use_bwd use_conv_sk use_engine use_io; do if [ $o = $F ]; then echo $R/build_ab/variant_$N.o; else echo $C/$o.o; fi; done)
rm -f $R/build_ab/variant_$N.o; ls -la $R/build_ab/libuse_hip_$N.so
