"""Time one optimisation step (train_step forward + backward + Adam) of NCSN++ large on the differentiable HIP operators.
Defaults = the reference's training configuration (configs/model/SGMSE_Large.yaml + configs/data/distort.yaml: batch 4, n_fft 1022,
hop 160, 512 frames, Adam lr 5e-4 weight_decay 1e-7, fp32).
usage: python scripts/train_step_bench.py [B] [num_frames] [steps] [n_fft] [hop]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from universal_speech_enhancement_amd.sgmse.model_wrapper import ScoreModel  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
NF = int(sys.argv[2]) if len(sys.argv) > 2 else 512
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
NFFT = int(sys.argv[4]) if len(sys.argv) > 4 else 1022
HOP = int(sys.argv[5]) if len(sys.argv) > 5 else 160
PREC = os.environ.get("TRAIN_PRECISION", "fp32")                            # fp32 | bf16 | fp16 (mixed precision)
for kv in os.environ.get("USE_OPTS", "").split(","):                      # e.g. USE_OPTS=conv_sk_max_px=4096
    if "=" in kv:
        from universal_speech_enhancement_amd.hip_engine import set_option
        set_option(kv.split("=")[0], int(kv.split("=")[1]))
torch.manual_seed(0)
m = ScoreModel(backbone="ncsnpplarge", sde="ouve", t_eps=3e-2, condition="noisy", n_fft=NFFT, hop_length=HOP, num_frames=NF,
               window="hann", sde_input="noisy", precision="fp32").cuda()
m.score_net.requires_grad_(True)
m.score_net.train_precision = PREC
opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=5e-4, weight_decay=1e-7)
L = (NF - 1) * HOP + 4000
clean = torch.randn(B, L, device="cuda") * 0.1
batch = {"clean": clean, "perturbed": clean + 0.05 * torch.randn_like(clean)}


def step():
    opt.zero_grad(set_to_none=True)
    loss = m.train_step(batch)
    loss.backward()
    opt.step()
    return loss


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / STEPS
print(f"train step [{PREC}] B={B} frames={NF} F={NFFT // 2 + 1 - 1}: {dt * 1e3:.1f} ms/step  ({B * NF / dt:.0f} frames/s)  loss {float(loss.detach()):.4g}  "
      f"peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
