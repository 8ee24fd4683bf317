#!/bin/bash
# kernel durations (rocprofv3) of the small-map convolution variants in the single-convolution harness
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_sk
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o sk -- python $R/scripts/gpu_conv_bench.py --variants 1,2,7 --cases "$1" --iters 20 --rounds 1 --no-check $2 > $OUT/stdout.log 2>&1
echo rc=$?; cat $OUT/stdout.log | tail -20
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$OUT/sk_kernel_trace.csv")))
agg=collections.OrderedDict()
for r in rows:
    n=r['Kernel_Name']
    if 'conv' not in n: continue
    k=(n[:70], int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), r['Grid_Size_Y'], r['Grid_Size_Z'])
    agg.setdefault(k, []).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in agg.items():
    v=sorted(v); print(k, len(v), 'median %.1f us min %.1f'%(v[len(v)//2], v[0]))
PY
