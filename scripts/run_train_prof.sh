#!/bin/bash
# training-step timing + per-kernel profile -> gpurun_out/train_*   usage: run_train_prof.sh [fp32|bf16|fp16]
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/.."
mkdir -p gpurun_out
export TRAIN_PRECISION=${1:-fp32}
python scripts/train_step_bench.py 4 512 5 > gpurun_out/train_bench.log 2>&1
export TMPDIR=/tmp
rm -rf gpurun_out/train_prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/train_prof -o train -- python scripts/train_step_bench.py 4 512 2 > gpurun_out/train_prof.log 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/train_prof/**/*kernel_stats.csv', recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
tot = sum(float(r['TotalDurationNs']) for r in rows)
with open('gpurun_out/train_kernel_stats.txt', 'w') as o:
    o.write(f"total kernel time per step {tot / 4e6:.1f} ms\n")
    for r in rows[:28]:
        o.write(f"{float(r['TotalDurationNs'])/tot*100:6.2f}%  calls {r['Calls']:>6}  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}\n")
PY
tail -1 gpurun_out/train_bench.log; cat gpurun_out/train_kernel_stats.txt
