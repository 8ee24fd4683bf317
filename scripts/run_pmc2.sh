#!/bin/bash
# memory-side PMC passes for one score evaluation
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmc2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/scripts/gpu_time_forward.py bf16 8 640 1"
i=0
for set in \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" \
  "TCC_REQ_sum TCC_BUSY_sum TCC_TAG_STALL_sum TCC_HIT_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_MISS_sum" \
  "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT -o pass$i -- $CMD > $OUT/pass$i.log 2>&1
  echo "pass $i rc=$?"
done
