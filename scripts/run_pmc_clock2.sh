#!/bin/bash
# clock + MFMA-busy (in cycles) of conv kernel variants on one case of the single-convolution harness: run_pmc_clock2.sh <out> "<variants>" "<case>" [iters]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_clock2}; mkdir -p $OUT; VARS=${2:-"4 6"}; CASE=${3:-"L0 conv0 128->128"}; IT=${4:-20}
cd /tmp && export TMPDIR=/tmp
for v in $VARS; do
  timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT -o v$v -- python $R/scripts/gpu_conv_bench.py --variants $v --cases "$CASE" --no-check --iters $IT --rounds 1 > $OUT/v$v.log 2>&1
  python - <<PY
import csv,glob
acc={}; n={}
for f in glob.glob("$OUT/**/v${v}_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_v" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]]=acc.get(r["Counter_Name"],0)+float(r["Counter_Value"]); n[r["Counter_Name"]]=n.get(r["Counter_Name"],0)+1
dur=[]
for f in glob.glob("$OUT/**/v${v}_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_v" in r["Kernel_Name"]: dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
dur=dur[2:]
g=acc["GRBM_GUI_ACTIVE"]/n["GRBM_GUI_ACTIVE"]/8; m=acc["SQ_VALU_MFMA_BUSY_CYCLES"]/n["SQ_VALU_MFMA_BUSY_CYCLES"]
t=sum(dur)/len(dur)/1e3
print(f"variant $v: {t:8.1f} us  clock {g/t/1e3:5.2f} GHz  mfma_busy(cycles) {m/(g*1024):.3f}  wait_any {acc['SQ_WAIT_ANY']/acc['SQ_WAVE_CYCLES']:.3f} wait_inst {acc['SQ_WAIT_INST_ANY']/acc['SQ_WAVE_CYCLES']:.3f} valu {acc['SQ_ACTIVE_INST_VALU']*4/n['SQ_ACTIVE_INST_VALU']/(g*1024):.3f} lds {acc['SQ_ACTIVE_INST_LDS']*4/n['SQ_ACTIVE_INST_LDS']/(g*1024):.3f}  launches {len(dur)}")
PY
done
