#!/bin/bash
# Energy of one convolution launch under sustained repetition: socket power (rocm-smi, once a second) x time per launch, per variant.
#   scripts/energy_per_launch.sh "<case substring>" "<variants, e.g. 4 10>" [iters]
CASE=${1:-L0 conv0 128->128}; VARS=${2:-4 10}; IT=${3:-16000}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R
for v in $VARS; do
  (python scripts/gpu_conv_bench.py --variants $v --no-check --rounds 1 --iters $IT --cases "$CASE" > /tmp/epl_$v.log 2>&1) & PID=$!
  sleep 4; W=(); C=()
  while kill -0 $PID 2>/dev/null; do
    P=$(rocm-smi --showpower 2>/dev/null | grep -oE "Power \(W\): [0-9.]+" | grep -oE "[0-9.]+$" | head -1)
    K=$(rocm-smi --showclocks 2>/dev/null | grep sclk | grep -oE "\([0-9]+Mhz\)" | head -1 | tr -dc 0-9)
    [ -n "$P" ] && W+=($P) && C+=($K); sleep 1
  done
  MS=$(grep -oE "v$v: +[0-9.]+ ms" /tmp/epl_$v.log | grep -oE "[0-9.]+ ms" | head -1 | tr -d ' ms')
  python3 - "$v" "$MS" "${W[*]}" "${C[*]}" <<'PY'
import sys
v, ms = sys.argv[1], float(sys.argv[2]); w = [float(x) for x in sys.argv[3].split()]; c = [float(x) for x in sys.argv[4].split()]
k = [i for i, x in enumerate(w) if x >= 0.9 * max(w)]
W = sum(w[i] for i in k) / len(k); C = sum(c[i] for i in k) / len(k)
print(f"variant {v}: {ms:.4f} ms per launch, {W:.0f} W at {C:.0f} MHz ({len(k)} samples) -> {W * ms * 1e-3:.3f} J per launch")
PY
done
