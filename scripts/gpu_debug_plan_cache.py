"""Bring-up: a second sampler call on the same plan (small shapes)."""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from universal_speech_enhancement_amd.hip_engine import HipScoreEngine
from universal_speech_enhancement_amd.testing import weights as tw, noise as tn
sd = tw.make_state_dict(1234, **tw.LARGE)
def seq(B, T, prec="bf16", corr="langevin", graph=True, N=2, n=3, score=False):
    eng = HipScoreEngine(precision=prec); eng.load_state_dict(sd)
    res = []
    y = torch.from_numpy(tn.complex_normal(5, f"y{B}", (B, 1, 512, T))).cuda() * 0.5
    for i in range(n):
        if score:
            out = eng.score(y, y, torch.full((B,), 0.5).cuda())
        else:
            eng.plan(B, T); eng.set_sampler(N, "reverse_diffusion", corr, 1, 0.5, 3e-2, use_graph=graph)
            out = eng.sample(y, noise=None, seed=3 + i)
        torch.cuda.synchronize()
        res.append(bool(torch.isfinite(torch.view_as_real(out)).all()))
    print(f"B={B} T={T} {prec} {corr} graph={graph} N={N} score={score}:", res, flush=True)
    eng.close()
seq(2, 64); seq(2, 64, graph=False); seq(2, 64, score=True); seq(8, 64); seq(2, 128); seq(1, 64); seq(2, 64, corr="none"); seq(2, 64, N=1); seq(2, 64, prec="fp32")
