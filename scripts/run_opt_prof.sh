#!/bin/bash
# per-kernel rocprofv3 statistics of one eager evaluation (bench --roofline-only style: gpu_time_forward with USE_HIP_PROFILE off is pipelined;
# here: 3 pipelined evaluations) for a list of USE_OPTS settings.  usage: run_opt_prof.sh <kernel-substring> <opts1> <opts2> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; PAT=$1; shift
cd /tmp && export TMPDIR=/tmp
for o in "$@"; do
  OUT=$R/gpurun_out/optprof_$(echo $o | tr '=,' '__'); mkdir -p $OUT
  USE_OPTS=$o timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o k -- python $R/scripts/gpu_time_forward.py bf16 8 640 3 > $OUT/stdout.log 2>&1
  python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$OUT/k_kernel_trace.csv")))
agg=collections.OrderedDict()
for r in rows:
    if "$PAT" not in r['Kernel_Name']: continue
    k=(r['Kernel_Name'][:40], int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']))
    agg.setdefault(k,[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print("== $o")
for k,v in agg.items():
    v=sorted(v); print("   ", k, len(v), "median %.1f us"%v[len(v)//2])
PY
done
