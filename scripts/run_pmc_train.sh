#!/bin/bash
# rocprofv3 PMC passes (counters only, with --kernel-trace) of one training step at a quarter of the reference's configuration.
# usage (on the GPU box): scripts/run_pmc_train.sh <outdir-under-gpurun_out> [fp32|bf16]
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_train}
export TRAIN_PRECISION=${2:-bf16}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/train_step_bench.py 2 512 1"
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
  "FETCH_SIZE GRBM_GUI_ACTIVE" \
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o pass$i -- $CMD > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
rm -f $OUT/*kernel_trace.csv
ls $OUT
