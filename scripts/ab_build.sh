#!/bin/bash
# same-box A/B of a compile-time switch: ab_build.sh <file.o> "<EXTRA flags for B>" [args of gpu_time_forward.py]
# (box-to-box variation is ~4 %, larger than most kernel changes)
OBJ=$1; FLAGS=$2; shift 2
cd universal_speech_enhancement_amd/csrc
for round in 1 2; do
  rm -f $OBJ; make >/dev/null 2>&1
  (cd ../..; echo "A(default): $(python scripts/gpu_time_forward.py "$@" 2>&1 | tail -1 | cut -c1-70)")
  rm -f $OBJ; make EXTRA="$FLAGS" >/dev/null 2>&1
  (cd ../..; echo "B($FLAGS): $(python scripts/gpu_time_forward.py "$@" 2>&1 | tail -1 | cut -c1-70)")
done
rm -f $OBJ; make >/dev/null 2>&1
