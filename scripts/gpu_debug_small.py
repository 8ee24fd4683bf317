import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from universal_speech_enhancement_amd.hip_engine import HipScoreEngine
from universal_speech_enhancement_amd.testing import weights as tw, noise as tn
arch = tw.SMALL12M
sd = tw.make_state_dict(4242, **arch)
x = torch.from_numpy(tn.complex_normal(17, "small_x", (2, 2, 512, 64))).cuda() * 0.5
t = torch.tensor([0.8, 0.1]).cuda()
taps = {}
for prec in ("fp32", "bf16"):
    e = HipScoreEngine(nf=96, ch_mult=arch["ch_mult"], num_res_blocks=1, precision=prec)
    e.load_state_dict(sd)
    out = e.forward(x[:, 0:1].contiguous(), x[:, 1:2].contiguous(), t)
    taps[prec] = {n: e.debug_tensor(n).cpu() for n in ("h_in", "down_out", "pre_attn", "post_attn", "h_last", "pyramid")}
    taps[prec]["out"] = torch.view_as_real(out).cpu()
for n in taps["fp32"]:
    a, b = taps["fp32"][n], taps["bf16"][n]
    print(f"{n:10s} shape {tuple(a.shape)}  relmax {float((a-b).abs().max()/a.abs().max()):.4f}  max {float(a.abs().max()):.3f}")
