#!/bin/bash
# Build an A/B variant of the library with extra -D flags on use_conv_v4.hip only:  scripts/build_variant.sh NAME [-DFLAG ...]
# -> build_ab/libuse_hip_NAME.so  (git-ignored; for scripts/ab_libs.py / scripts/profile.sh ab|trace)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/universal_speech_enhancement_amd/csrc; N=$1; shift
mkdir -p $R/build_ab
make -s -C $C -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c $C/use_conv_v4.hip -o $R/build_ab/use_conv_v4_$N.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "conv_v4_kernelIDF16bS0_Li32ELb1E\|error" | grep -i "error\|VGPRs:\|Spill\|LDS Size" | head -8 || true
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/build_ab/libuse_hip_$N.so $C/use_kernels.o $C/use_conv_v2.o $R/build_ab/use_conv_v4_$N.o $C/use_attn.o $C/use_bwd.o $C/use_conv_sk.o $C/use_engine.o $C/use_io.o
rm -f $R/build_ab/use_conv_v4_$N.o; ls -la $R/build_ab/libuse_hip_$N.so
