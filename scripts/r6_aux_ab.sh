#!/bin/bash
# round 6: cache policy of conv_v4's output stores by map size (maps below 512 rows fit the memory-side cache), same-box end-to-end A/B
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for rep in 1 2; do for l in universal_speech_enhancement_amd/libuse_hip.so build_ab/libuse_hip_aux_s0.so build_ab/libuse_hip_aux_s16.so build_ab/libuse_hip_aux_s2.so build_ab/libuse_hip_aux_all0.so build_ab/libuse_hip_aux_res0.so; do
  printf "%-44s " "$l"; USE_HIP_LIB=$R/$l python scripts/gpu_time_forward.py bf16 8 640 200 2>&1 | tail -1 | cut -c1-60
done; done
