#!/bin/bash
# PMC passes (counters only) of the fp32 weight-gradient kernel launched back to back (scripts/wgrad_steady.py shapes)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_wgrad; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/wg1.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from universal_speech_enhancement_amd import training_ops as T
dy = torch.randn(4, 512, 512, 128, device="cuda") * 0.5; x = torch.randn(4, 512, 512, 128, device="cuda")
for _ in range(6): T.conv_wgrad(dy, x)
torch.cuda.synchronize()
PY
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
  "FETCH_SIZE GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o pass$i -- python /tmp/wg1.py > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
rm -f $OUT/*kernel_trace.csv
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob("$OUT/*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        if "wgrad_tile_kernel" not in row["Kernel_Name"]: continue
        a = acc[row["Counter_Name"]]; a[0] += float(row["Counter_Value"]); a[1] += 1
avg = {k: v[0] / v[1] for k, v in acc.items()}
g = avg.get("GRBM_GUI_ACTIVE", 0)
for k, v in sorted(avg.items()): print(f"{k:28s} {v:16.0f}")
if g:
    print("mfma_busy_frac", avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 128))
    print("valu_active_frac", avg["SQ_ACTIVE_INST_VALU"] / (g * 64), "lds_active_frac", avg["SQ_LDS_IDX_ACTIVE"] / (g * 32))
print("wait_any", avg["SQ_WAIT_ANY"] / avg["SQ_WAVE_CYCLES"], "wait_inst_any", avg["SQ_WAIT_INST_ANY"] / avg["SQ_WAVE_CYCLES"], "wait_inst_lds", avg["SQ_WAIT_INST_LDS"] / avg["SQ_WAVE_CYCLES"])
print("bank_conflict/lds_active", avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"])
PY
