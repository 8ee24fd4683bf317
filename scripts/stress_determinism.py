"""40 score evaluations of the configs[1] shape (two sub-batch streams) must be bit-identical (integer-atomic GroupNorm totals, fixed
summation orders): python scripts/stress_determinism.py"""
import os, sys, zlib
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
from universal_speech_enhancement_amd.hip_engine import HipScoreEngine
from universal_speech_enhancement_amd.testing import weights, noise as tn
sd = weights.make_state_dict(1234)
e = HipScoreEngine(precision="bf16"); e.load_state_dict(sd)
B, Tp = 8, 640
x = torch.from_numpy(tn.complex_normal(3, "x", (B, 1, 512, Tp))).cuda()
y = torch.from_numpy(tn.complex_normal(3, "y", (B, 1, 512, Tp))).cuda() * 0.5
t = torch.full((B,), 0.5, device="cuda")
e.plan(B, Tp)
crcs = set()
for i in range(40):
    s = e.score(x, y, t)
    torch.cuda.synchronize()
    crcs.add(zlib.crc32(torch.view_as_real(s).cpu().numpy().tobytes()))
print("distinct results over 40 evaluations:", len(crcs), "finite:", bool(torch.isfinite(torch.view_as_real(s)).all()))
