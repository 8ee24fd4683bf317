#!/bin/bash
# round 6 measurements, set A (one box):
#  1. matrix pipe alone, bf16 vs f16 on random data, sustained under the power cap (where does fp16's 4 % go?)
#  2. joules per launch of the small-map convolutions (conv_v2 at 64x80 / 32x40, conv_sk at 16x20 / 8x10) next to conv_v4's: how much of an
#     evaluation's energy they are (what a better small-map kernel could buy)
#  3. one conv_v4 launch in bf16 and fp16 storage, sustained
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for wps in 2 1; do for dt in 0 1 0 1; do
  python scripts/smi_probe.py --period 0.5 -- scripts/microbench/mfma_dtype_power $dt 6 $wps > /tmp/mp.log 2>&1
  python3 - <<'PY'
import json
L = open("/tmp/mp.log").read().strip().splitlines()
j = json.loads(L[-1]); print(L[-2], "|", j.get("socket_W"), "W", j.get("current_gfxclk"), "MHz ppt", (j.get("limit_reasons") or {}).get("ppt_pwr"))
PY
done; done
LIB=universal_speech_enhancement_amd/libuse_hip.so
echo "== conv_v4 sustained, bf16 (dtype 1) vs fp16 (dtype 2), 3-item launches"
for c in "L0 conv0 128->128" "L0 conv1 128->128 +res" "L1 conv0 cat384->128"; do for dt in 1 2 1 2; do
  echo -n "dtype $dt $c: "; EA_OPTS="--dtype $dt --batch 3" scripts/energy_ablation.sh "$c" "$LIB" 6000
done; done
echo "== small maps, 3-item launches (variant 0 = the dispatcher's choice)"
for c in "L3 conv0 256->256" "L3 conv0 cat512->256" "L4 conv0 256->256" "L4 conv0 cat512->256" "L5 conv0 256->256" "L6 conv0 256->256"; do
  echo -n "$c: "; VARIANT=0 EA_OPTS="--batch 3" scripts/energy_ablation.sh "$c" "$LIB" 20000
done
