#!/bin/bash
# same-box comparison of several compile-time variants of one object: ab_multi.sh <file.o> "<flags1>" "<flags2>" ...
# environment: ARGS (arguments of gpu_time_forward.py, default "bf16 8 640 4")
OBJ=$1; shift
ARGS=${ARGS:-"bf16 8 640 4"}
cd universal_speech_enhancement_amd/csrc
for round in 1 2; do
  for FLAGS in "" "$@"; do
    rm -f $OBJ; make EXTRA="$FLAGS" >/dev/null 2>&1
    (cd ../..; echo "[$FLAGS] $(python scripts/gpu_time_forward.py $ARGS 2>&1 | tail -1 | cut -c1-48)")
  done
done
rm -f $OBJ; make >/dev/null 2>&1
