import sys, os
sys.path.insert(0, os.getcwd())
import torch
from universal_speech_enhancement_amd.hip_engine import HipScoreEngine
from universal_speech_enhancement_amd.testing import weights as tw, noise as tn
for prec in ("bf16",):
    eng = HipScoreEngine(precision=prec); eng.load_state_dict(tw.make_state_dict(1234, **tw.LARGE))
    for (B, T) in ((1, 2048), (2, 4096), (3, 704)):
        x = torch.from_numpy(tn.complex_normal(1, "x", (B, 1, 512, T))).cuda() * 0.5
        y = torch.from_numpy(tn.complex_normal(1, "y", (B, 1, 512, T))).cuda() * 0.5
        t = torch.full((B,), 0.5).cuda()
        out = eng.score(x, y, t); torch.cuda.synchronize()
        # batch independence: item 0 alone gives the same result as item 0 in the batch
        o1 = eng.score(x[:1].contiguous(), y[:1].contiguous(), t[:1]); torch.cuda.synchronize()
        print(prec, B, T, "finite", bool(torch.isfinite(torch.view_as_real(out)).all()), "ws GB %.1f" % (eng.workspace_bytes()/1e9),
              "item0 same alone:", bool(torch.equal(out[:1], o1)))
