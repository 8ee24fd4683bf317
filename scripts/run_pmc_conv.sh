#!/bin/bash
# rocprofv3 PMC passes (counters only) over the single-convolution harness. usage: run_pmc_conv.sh <outdir> <variants> <case-substring>
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc_conv}; V=${2:-4,5}; CASE=${3:-"L0 conv0 128->128"}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
  "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM" \
  "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" \
  "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o pass$i -- python $R/scripts/gpu_conv_bench.py --variants $V --cases "$CASE" --no-check --iters 3 --rounds 1 > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
python $R/scripts/pmc_summary.py $OUT conv_v > $OUT/summary.txt 2>&1
