#!/bin/bash
# clock + MFMA-busy of the single-convolution harness under ablation bits (USE_HIP_ABLATE build of conv_v5): one PMC pass each
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_clock}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for d in 0 1 2 4 8 15; do
  USE_HIP_DBG=$d timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $OUT -o dbg$d -- python $R/scripts/gpu_conv_bench.py --variants 5 --cases "L0 conv0 128->128" --no-check --iters 3 --rounds 1 > $OUT/dbg$d.log 2>&1
  python - <<PY
import csv,glob
acc={}; n={}
for f in glob.glob("$OUT/**/dbg${d}_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_v5" not in r["Kernel_Name"]: continue
        acc[r["Counter_Name"]]=acc.get(r["Counter_Name"],0)+float(r["Counter_Value"]); n[r["Counter_Name"]]=n.get(r["Counter_Name"],0)+1
dur=[]
for f in glob.glob("$OUT/**/dbg${d}_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_v5" in r["Kernel_Name"]: dur.append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
g=acc["GRBM_GUI_ACTIVE"]/n["GRBM_GUI_ACTIVE"]/8; m=acc["SQ_VALU_MFMA_BUSY_CYCLES"]/n["SQ_VALU_MFMA_BUSY_CYCLES"]
t=sum(dur)/len(dur)/1e3
print(f"dbg=$d: {t:8.1f} us  clock {g/t/1e3:5.2f} GHz  mfma_busy {m/(g*1024):.3f}  wait_any {acc['SQ_WAIT_ANY']/acc['SQ_WAVE_CYCLES']:.3f} wait_inst {acc['SQ_WAIT_INST_ANY']/acc['SQ_WAVE_CYCLES']:.3f} valu_active {acc['SQ_ACTIVE_INST_VALU']*4/n['SQ_ACTIVE_INST_VALU']/(g*1024):.3f}")
PY
done
