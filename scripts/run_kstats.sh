#!/bin/bash
# rocprofv3 kernel stats of a few score evaluations (quick per-kernel view). usage: run_kstats.sh <tag>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/ks_${1:-x}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o ks -- python $R/scripts/gpu_time_forward.py bf16 8 640 3 > $OUT/stdout.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/ks_kernel_stats.csv")))
n=5.0  # 2 warm + 3 timed
for r in rows[:16]:
    print(f"{float(r['TotalDurationNs'])/1e6/n:8.3f} ms/score calls/score={int(r['Calls'])/n:6.1f} avg={float(r['AverageNs'])/1e3:8.1f} us max={float(r['MaxNs'])/1e3:8.1f}  {r['Name'][:80]}")
PY
