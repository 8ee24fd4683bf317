#!/bin/bash
# conv_v4w same-box A/B: correctness vs conv_v4, sustained single-convolution runs (time, power, joules per launch), one score evaluation off / on
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
L=universal_speech_enhancement_amd/libuse_hip.so
echo "== check (ipw 2, B = 4 and 3; fp16)"
python scripts/gpu_conv_bench.py --variants 4,5 --iters 3 --rounds 1 --cases "L0 conv0 128->128,L0 conv1 128->128 +res,L0 conv0 cat256->128,L1 conv0 128->128 noact,L2 conv0 cat512->256" --opt conv_v4w_ipw=2 | cut -c1-200
python scripts/gpu_conv_bench.py --variants 4,5 --iters 3 --rounds 1 --batch 3 --cases "L1 conv0 128->128,L2 conv1 256->256 +res" --opt conv_v4w_ipw=2 | cut -c1-200
python scripts/gpu_conv_bench.py --variants 4,5 --iters 3 --rounds 1 --dtype 2 --cases "L1 conv1 128->128 +res" --opt conv_v4w_ipw=2 | cut -c1-200
for c in "L0 conv0 128->128" "L0 conv1 128->128 +res"; do
  echo "== sustained: $c  (v4 | v4w ipw 1 | v4w ipw 2 | v4w ipw 2 with ds_write_b32 staging)"
  VARIANT=4 scripts/energy_ablation.sh "$c" $L 8000
  VARIANT=5 EA_OPTS="--opt conv_v4w_ipw=1" scripts/energy_ablation.sh "$c" $L 8000
  VARIANT=5 EA_OPTS="--opt conv_v4w_ipw=2" scripts/energy_ablation.sh "$c" $L 8000
  VARIANT=5 EA_OPTS="--opt conv_v4w_ipw=2" scripts/energy_ablation.sh "$c" build_ab/libuse_hip_plainstg.so 8000
done
for rep in 1 2; do
for o in "conv_v4w=0" "conv_v4w=1"; do
  echo "== e2e $o"; USE_OPTS="$o" python scripts/gpu_time_forward.py bf16 8 640 200 2>&1 | tail -1
done; done
