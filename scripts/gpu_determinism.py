import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from universal_speech_enhancement_amd.hip_engine import HipScoreEngine, set_option
from universal_speech_enhancement_amd.testing import weights as tw, noise as tn
for o in sys.argv[1:]:
    k, v = o.split("="); set_option(k, int(v))
sd = tw.make_state_dict(1234, **tw.LARGE)
for prec in ("fp32", "bf16"):
    e = HipScoreEngine(precision=prec); e.load_state_dict(sd)
    for B, T in ((1, 64), (3, 64), (8, 128)):
        x = torch.from_numpy(tn.complex_normal(1, "x", (B, 1, 512, T))).cuda() * 0.5
        y = torch.from_numpy(tn.complex_normal(1, "y", (B, 1, 512, T))).cuda() * 0.5
        t = torch.full((B,), 0.5).cuda()
        outs = [e.score(x, y, t).clone() for _ in range(4)]
        print(prec, B, T, "equal:", [bool(torch.equal(outs[0], o)) for o in outs[1:]], "maxdiff", max(float((outs[0]-o).abs().max()) for o in outs[1:]))
    e.close()
