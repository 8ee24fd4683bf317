"""GPU bring-up helper: time one score evaluation at a given shape (not a test)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from universal_speech_enhancement_amd.hip_engine import HipScoreEngine
from universal_speech_enhancement_amd.testing import weights as tw
from universal_speech_enhancement_amd.testing import noise as tn

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = int(sys.argv[3]) if len(sys.argv) > 3 else 640
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
from universal_speech_enhancement_amd.hip_engine import set_option
for env, opt in (("USE_SUBBATCH", "subbatch"), ("USE_STAGGER", "stagger_level")):      # A/B of the scheduling knobs
    if os.environ.get(env):
        set_option(opt, int(os.environ[env]))
for kv in filter(None, os.environ.get("USE_OPTS", "").split(",")):                       # generic: USE_OPTS="gn_inline=0,subbatch=2"
    set_option(kv.split("=")[0], int(kv.split("=")[1]))
eng = HipScoreEngine(precision=prec)
eng.load_state_dict(tw.make_state_dict(1234, **tw.LARGE))
x = torch.from_numpy(tn.complex_normal(1, "x", (B, 1, 512, T))).cuda() * 0.5
y = torch.from_numpy(tn.complex_normal(1, "y", (B, 1, 512, T))).cuda() * 0.5
t = torch.full((B,), 0.5).cuda()
out = eng.score(x, y, t); torch.cuda.synchronize()
print("workspace %.2f GB, flops/score %.2f TF" % (eng.workspace_bytes() / 1e9, eng.flops_per_score() / 1e12))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    out = eng.score(x, y, t)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(f"[{prec}] B={B} T'={T}: {ms:.2f} ms / score  -> {eng.flops_per_score() / ms / 1e9:.1f} TFLOP/s, {B*T/ms*1e3:.0f} padded-frame*NFE/s, finite={torch.isfinite(torch.view_as_real(out)).all().item()}")
if os.environ.get("USE_HIP_PROFILE_VERBOSE"):
    eng.profile_score(x, y, t)
    ms, fl, by, n, tot = eng.profile_score(x, y, t)
    print(f"conv launches {n}: {ms:.2f} ms of {tot:.2f} ms total, {fl/ms/1e9:.1f} TFLOP/s")
