"""Steady-state rate of the weight-gradient kernels: the same launch back to back (the chip settles at its power-managed clock), against
the rate the kernel shows inside a training step (between bandwidth-bound kernels, at boost clock)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from universal_speech_enhancement_amd import training_ops as T  # noqa: E402

B, H, W, C = 4, 512, 512, 128
for dt in (torch.bfloat16, torch.float32):
    dy = (torch.randn(B, H, W, C, device="cuda") * 0.5).to(dt)
    x = torch.randn(B, H, W, C, device="cuda").to(dt)
    for n in (1, 5, 50, 200):
        T.conv_wgrad(dy, x); torch.cuda.synchronize()
        time.sleep(0.5)                                          # let the chip cool to idle clocks first
        t0 = time.perf_counter()
        for _ in range(n):
            T.conv_wgrad(dy, x)
        torch.cuda.synchronize()
        dtm = (time.perf_counter() - t0) / n
        fl = 2.0 * B * H * W * C * C * 9
        print(f"{str(dt):16s} {n:4d} launches back to back: {dtm * 1e6:8.1f} us each = {fl / dtm / 1e12:7.1f} TFLOP/s (incl. the slice reduction)")
