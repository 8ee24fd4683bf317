#!/bin/bash
# same-box A/B of library builds: per-kernel rocprofv3 statistics (eager evaluation: roofline-only bench) + end-to-end time
# usage: ab_kernel.sh <kernel-name-substring> <libA.so> <libB.so> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
for L in "$@"; do
  OUT=$R/gpurun_out/abk_$(basename $L .so); mkdir -p $OUT
  USE_HIP_LIB=$R/$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $R/bench.py --roofline-only > $OUT/stdout.log 2>&1
  echo "== $L"; grep "$PAT" $OUT/k_kernel_stats.csv | awk -F, '{printf "   calls %s avg %.1f us total %.2f ms  %s\n", $2, $4/1e3, $3/1e6, substr($1,1,60)}'
done
cd $R
for r in 1 2; do for L in "$@"; do echo "$L $(USE_HIP_LIB=$L python scripts/gpu_time_forward.py bf16 8 640 5 2>&1 | tail -1 | cut -c1-45)"; done; done
