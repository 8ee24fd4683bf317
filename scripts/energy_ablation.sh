#!/bin/bash
# Where the joules of one convolution launch go: the single-convolution harness repeated for ~6 s per library build (timing-only
# ablation builds of conv_v4: scripts/build_variant.sh abl<N> -DV4_ABL=<N>), socket power / PPT-violation share / clock sampled
# by scripts/smi_probe.py.   [VARIANT=2 EA_OPTS="--opt name=value"] scripts/energy_ablation.sh "<case>" "<lib1> <lib2> ..." [iters]
CASE=${1:-L0 conv0 128->128}; LIBS=${2:-universal_speech_enhancement_amd/libuse_hip.so}; IT=${3:-14000}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for lib in $LIBS; do
  USE_HIP_LIB=$R/$lib python scripts/smi_probe.py --period 0.5 -- python scripts/gpu_conv_bench.py --variants ${VARIANT:-4} ${EA_OPTS:-} --no-check --rounds 1 --iters $IT --cases "$CASE" > /tmp/ea.log 2>&1
  MS=$(grep -oE "v[0-9]+: +[0-9.]+ ms" /tmp/ea.log | grep -oE "[0-9]+\.[0-9]+" | head -1)
  J=$(tail -1 /tmp/ea.log)
  python3 - "$lib" "$MS" "$J" <<'PY'
import sys, json
lib, ms = sys.argv[1], float(sys.argv[2] or 0); j = json.loads(sys.argv[3])
lr = j.get("limit_reasons", {})
print(f"{lib.split('/')[-1]:34s} {ms:.4f} ms  {j.get('socket_W', 0):6.0f} W  {j.get('current_gfxclk', 0):5.0f} MHz  ppt {lr.get('ppt_pwr')}  -> {j.get('socket_W', 0) * ms * 1e-3:.4f} J per launch  (dyn over 240 W idle: {(j.get('socket_W', 0) - 240) * ms * 1e-3:.4f} J)")
PY
done
