#!/bin/bash
# same-box kernel statistics of one configuration per USE_OPTS string: run_v7_kstats.sh "<opts A>" "<opts B>" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for O in "$@"; do
  i=$((i+1)); OUT=$R/gpurun_out/kst_$i; mkdir -p $OUT
  USE_OPTS="$O" timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o k -- python $R/scripts/gpu_time_forward.py bf16 8 640 5 > $OUT/stdout.log 2>&1
  echo "== $O"; tail -1 $OUT/stdout.log | cut -c1-60
  awk -F, 'NR>1{gsub(/"/,"",$1); printf "   %-60s calls %6s avg %8.1f us total %8.2f ms\n", substr($1,1,60), $2, $4/1e3, $3/1e6}' $OUT/k_kernel_stats.csv | head -14
  rm -f $OUT/*kernel_trace.csv
done
