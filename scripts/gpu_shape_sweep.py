"""GPU bring-up helper: odd batch sizes / frame counts x sub-batch counts; checks finiteness and bit-equality across splits."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from universal_speech_enhancement_amd.hip_engine import HipScoreEngine, set_option
from universal_speech_enhancement_amd.testing import weights as tw, noise as tn
sd = tw.make_state_dict(1234, **tw.LARGE)
for prec in ("bf16", "fp32"):
    outs = {}
    for nsub in (1, 2, 3):
        set_option("subbatch", nsub)
        e = HipScoreEngine(precision=prec); e.load_state_dict(sd)
        for B, T in ((1, 64), (3, 128), (5, 64), (7, 64), (2, 320)):
            x = torch.from_numpy(tn.complex_normal(B, "x", (B, 1, 512, T))).cuda() * 0.5
            y = torch.from_numpy(tn.complex_normal(B, "y", (B, 1, 512, T))).cuda() * 0.5
            t = torch.linspace(0.9, 0.05, B).cuda()
            o = e.score(x, y, t)
            assert torch.isfinite(torch.view_as_real(o)).all(), (prec, nsub, B, T)
            k = (B, T)
            if k in outs: assert torch.equal(outs[k], o), (prec, nsub, B, T)
            else: outs[k] = o.clone()
        e.close()
    print(prec, "ok")
set_option("subbatch", -1)
