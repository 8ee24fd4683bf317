#!/bin/bash
# bring-up: socket power / shader clock while (a) the fp32 weight-gradient kernel runs back to back, (b) the fp32 training step loops
R=${GRAFT_REPO_ROOT:-$(pwd)}
probe() {
  while kill -0 $1 2>/dev/null; do
    P=$(rocm-smi --showpower 2>/dev/null | grep -oE "Power \(W\): [0-9.]+" | grep -oE "[0-9.]+$")
    C=$(rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk|fclk" | grep -oE "[a-z]+ clock level: [0-9]+: \([0-9]+Mhz\)" | tr '\n' ' ')
    echo "power $P W  $C"
    sleep 1
  done
}
echo "== wgrad fp32 back to back"
(python - <<PY > $R/gpurun_out/pp_a.log 2>&1
import sys, torch, time
sys.path.insert(0, "$R")
from universal_speech_enhancement_amd import training_ops as T
dy = torch.randn(4, 512, 512, 128, device="cuda") * 0.5; x = torch.randn(4, 512, 512, 128, device="cuda")
torch.cuda.synchronize(); t0 = time.time()
while time.time() - t0 < 12:
    for _ in range(20): T.conv_wgrad(dy, x)
    torch.cuda.synchronize()
PY
) &
probe $!
echo "== fp32 training steps"
(python $R/scripts/train_step_bench.py 4 512 40 > $R/gpurun_out/pp_b.log 2>&1) &
probe $!
tail -1 $R/gpurun_out/pp_b.log
