#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 4 7; do
OUT=$R/gpurun_out/prof_skabl$d; mkdir -p $OUT
USE_HIP_DBG=$d USE_HIP_LIB=$R/scripts/ab/libuse_hip_abl.so timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT -o sk -- python $R/scripts/gpu_conv_bench.py --variants 7 --cases "L4 conv0 256,L5 conv0 256,L6 conv0 256" --iters 20 --rounds 1 --no-check > $OUT/stdout.log 2>&1
python - <<PY
import csv, collections
rows=list(csv.DictReader(open("$OUT/sk_kernel_trace.csv")))
agg=collections.OrderedDict()
for r in rows:
    n=r['Kernel_Name']
    if 'conv_sk' not in n: continue
    k=(int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), r['Grid_Size_Y'])
    agg.setdefault(k, []).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
print("dbg=$d", {k: round(sorted(v)[len(v)//2],1) for k,v in agg.items()})
PY
done
