#!/bin/bash
# rocprofv3 kernel statistics of a few score evaluations at batch $1 (default 1): which kernels make the small-batch latency
R=${GRAFT_REPO_ROOT:-$(pwd)}; B=${1:-1}
OUT=$R/gpurun_out/ks_b$B; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- python $R/scripts/gpu_time_forward.py bf16 $B 640 5 > $OUT/stdout.log 2>&1
tail -1 $OUT/stdout.log
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$OUT/ks_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows); n = 7.0
print("kernel time per evaluation %.2f ms" % (tot / 1e6 / n))
for r in rows[:22]:
    print("%6.2f%%  calls/eval %6.1f  avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / tot * 100, int(r["Calls"]) / n, float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY
