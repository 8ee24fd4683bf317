"""Steady-state rate of the weight-gradient kernels: the same launch back to back, with random / constant / zero operands and with
operands that fit the 256 MB last-level cache - against the rate the kernel shows inside a training step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from universal_speech_enhancement_amd import training_ops as T  # noqa: E402


def run(dt, B, H, W, C, fill, n=100):
    if fill == "randn":
        dy = (torch.randn(B, H, W, C, device="cuda") * 0.5).to(dt); x = torch.randn(B, H, W, C, device="cuda").to(dt)
    elif fill == "ones":
        dy = torch.full((B, H, W, C), 0.5, device="cuda").to(dt); x = torch.ones(B, H, W, C, device="cuda").to(dt)
    else:
        dy = torch.zeros(B, H, W, C, device="cuda", dtype=dt); x = torch.zeros(B, H, W, C, device="cuda", dtype=dt)
    for _ in range(3):
        T.conv_wgrad(dy, x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        T.conv_wgrad(dy, x)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    fl = 2.0 * B * H * W * C * C * 9
    print(f"{str(dt):15s} B={B} {H}x{W} C={C} {fill:6s}: {ms * 1e3:8.1f} us per call = {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)


for dt in (torch.bfloat16, torch.float32):
    for fill in ("randn", "ones", "zeros"):
        run(dt, 4, 512, 512, 128, fill)
    run(dt, 1, 512, 512, 128, "randn")
    run(dt, 1, 256, 256, 128, "randn", n=400)
