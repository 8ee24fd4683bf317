import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from universal_speech_enhancement_amd.hip_engine import HipScoreEngine, set_option
from universal_speech_enhancement_amd.testing import weights as tw, noise as tn
sd = tw.make_state_dict(1234, **tw.LARGE)
x = torch.from_numpy(tn.complex_normal(5, "x", (5, 1, 512, 64))).cuda() * 0.5
y = torch.from_numpy(tn.complex_normal(5, "y", (5, 1, 512, 64))).cuda() * 0.5
t = torch.linspace(0.9, 0.1, 5).cuda()
for gi in (20480, 0, 400000):
    set_option("gn_inline", gi)
    outs = {}
    for n in (1, 2):
        set_option("subbatch", n)
        e = HipScoreEngine(precision="bf16"); e.load_state_dict(sd)
        a = e.score(x, y, t).clone(); b = e.score(x, y, t).clone()
        outs[n] = a
        print(f"gn_inline={gi} subbatch={n}: repeat equal {torch.equal(a, b)}")
        e.close()
    d = (outs[1] - outs[2]).abs()
    print("  1 vs 2: equal", torch.equal(outs[1], outs[2]), "per-item maxdiff", [float(d[i].max()) for i in range(5)])
