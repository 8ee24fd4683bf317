#!/bin/bash
# Collect rocprofv3 PMC passes (counters only, with --kernel-trace) for one score evaluation at the cfg2 shape.
# usage (on the GPU box): scripts/run_pmc.sh <outdir-under-gpurun_out> [precision B T]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc}
PREC=${2:-bf16}; B=${3:-8}; T=${4:-640}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/gpu_time_forward.py $PREC $B $T 1"
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o pass$i -- $CMD > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
ls $OUT
