"""Same-process A/B of a use_set_option knob: one score evaluation at configs[1] with value A and value B -- bit-wise comparison of
the outputs and interleaved timing.   python scripts/gpu_ab_option.py <option> <A> <B> [precision]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from universal_speech_enhancement_amd.hip_engine import HipScoreEngine, set_option
from universal_speech_enhancement_amd.testing import noise as tn
from universal_speech_enhancement_amd.testing import weights as tw

opt, A, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
x = torch.from_numpy(tn.complex_normal(1, "x", (8, 1, 512, 640))).cuda() * 0.5
y = torch.from_numpy(tn.complex_normal(1, "y", (8, 1, 512, 640))).cuda() * 0.5
t = torch.full((8,), 0.5).cuda()
outs = {}
for val in (A, B, A, B):
    set_option(opt, val)
    eng = HipScoreEngine(precision=prec)
    eng.load_state_dict(tw.make_state_dict(1234, **tw.LARGE))
    out = eng.score(x, y, t); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = eng.score(x, y, t)
    e1.record(); torch.cuda.synchronize()
    print(f"{opt}={val}: {e0.elapsed_time(e1) / 5:.2f} ms / score", flush=True)
    outs.setdefault(val, out.clone())
    eng.close()
d = (torch.view_as_real(outs[A]) - torch.view_as_real(outs[B])).abs().max().item()
print(f"max |out(A) - out(B)| = {d:.3g}  (max |out| = {torch.view_as_real(outs[A]).abs().max().item():.3g})")
