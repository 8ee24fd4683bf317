#!/bin/bash
# round-5 bring-up: what the SMI reports under load + the timeline of one graph-replayed sampler call
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
OUT=$R/gpurun_out/explore; mkdir -p $OUT
amd-smi metric --help > $OUT/amd_smi_metric_help.txt 2>&1
amd-smi monitor --help > $OUT/amd_smi_monitor_help.txt 2>&1
amd-smi static -g 0 --limit > $OUT/amd_smi_static_limit.txt 2>&1
python scripts/smi_probe.py --dump --period 1.0 -- python scripts/gpu_time_forward.py bf16 8 640 400 > $OUT/smi_probe_eval.txt 2>&1
echo "smi rc=$?"; tail -2 $OUT/smi_probe_eval.txt | cut -c1-1500
(python scripts/gpu_time_forward.py bf16 8 640 300 > /dev/null 2>&1) & PID=$!
sleep 12
amd-smi metric -g 0 > $OUT/amd_smi_metric_under_load.txt 2>&1
amd-smi monitor -g 0 --violation > $OUT/amd_smi_monitor_violation.txt 2>&1 &
MP=$!; sleep 4; kill $MP 2>/dev/null
wait $PID
cd /tmp && export TMPDIR=/tmp; mkdir -p $OUT/prof
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o bench -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-power-probe > $OUT/prof/stdout.log 2>&1
echo "prof rc=$?"
TR=$(ls $OUT/prof/*kernel_trace.csv | head -1)
python $R/scripts/trace_gaps.py $TR > $OUT/trace_gaps.txt 2>&1; cat $OUT/trace_gaps.txt
# keep a few steady-state evaluations of the trace (small) for offline analysis
python - "$TR" "$OUT/one_call_trace.csv.gz" <<'PY'
import csv, sys, gzip
rows = list(csv.DictReader(open(sys.argv[1])))
pri = [int(r['Start_Timestamp']) for r in rows if 'prior_kernel' in r['Kernel_Name']]
t0 = max(pri)
keep = [r for r in rows if int(r['Start_Timestamp']) >= t0]
keep.sort(key=lambda r: int(r['Start_Timestamp']))
n = len(keep); a = n // 2; b = min(n, a + 6 * 600)
cols = ['Kernel_Name', 'Start_Timestamp', 'End_Timestamp', 'Grid_Size_X', 'Grid_Size_Y', 'Grid_Size_Z', 'Workgroup_Size_X', 'Queue_Id', 'Stream_Id']
cols = [c for c in cols if c in keep[0]]
with gzip.open(sys.argv[2], 'wt') as f:
    w = csv.writer(f); w.writerow(cols)
    for r in keep[a:b]:
        row = [r[c] for c in cols]; row[0] = row[0][:60]; w.writerow(row)
print("kept", b - a, "rows; columns", list(keep[0].keys()))
PY
rm -f $OUT/prof/*kernel_trace.csv
ls -la $OUT
