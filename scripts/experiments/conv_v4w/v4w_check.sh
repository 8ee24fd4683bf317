#!/bin/bash
# conv_v4w bring-up: harness comparison with conv_v4 (outputs must be bit-identical, GroupNorm totals to fp32 rounding), ipw = 1 / 2 / 4
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
CASES="L0 conv0 128->128,L0 conv1 128->128 +res,L0 conv0 cat256->128,L1 conv0 128->128 noact,L1 conv0 256->256,L2 conv0 256->256,L2 conv0 cat512->256"
for ipw in 1 2 4; do
  echo "== ipw $ipw (B=4)"; timeout 600 python scripts/gpu_conv_bench.py --variants 4,5 --iters 20 --rounds 2 --cases "$CASES" --opt conv_v4w_ipw=$ipw
done
echo "== ipw 2, B=3 (odd: last workgroup walks one item)"; timeout 600 python scripts/gpu_conv_bench.py --variants 4,5 --iters 5 --rounds 1 --batch 3 --cases "L1 conv0 128->128,L2 conv1 256->256 +res" --opt conv_v4w_ipw=2
echo "== fp16"; timeout 600 python scripts/gpu_conv_bench.py --variants 4,5 --iters 5 --rounds 1 --dtype 2 --cases "L1 conv0 128->128,L1 conv1 128->128 +res" --opt conv_v4w_ipw=2
