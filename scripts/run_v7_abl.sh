#!/bin/bash
# bring-up: ablations of conv_v7's epilogue in the trace build (bits 16..: no stores / no statistics / no epilogue / staggered start)
export USE_HIP_LIB=$PWD/universal_speech_enhancement_amd/libuse_hip_trace.so
for A in ${ABL:-100 65636 131172 196708 262244 524388}; do
  echo "== dbg $A"; USE_HIP_TRACE=$A python scripts/gpu_conv_trace.py ${VAR:-8} "${1:-L0 conv0 128->128}" 2>&1 | grep -E "ms|G0" | sed -n '30,50p;$p'
done
