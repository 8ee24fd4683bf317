#!/bin/bash
# bring-up: time conv_v3 with parts of the helper's work removed (results are wrong; timing only)
cd universal_speech_enhancement_amd/csrc
for abl in 0 8 32 128 256; do
  rm -f use_conv_v3.o; make EXTRA=-DV3_ABL=$abl >/dev/null 2>&1
  (cd ../..; echo "ABL=$abl"; USE_HIP_V3=1 python scripts/gpu_time_forward.py bf16 8 640 2 2>&1 | tail -1)
done
rm -f use_conv_v3.o; make >/dev/null 2>&1
