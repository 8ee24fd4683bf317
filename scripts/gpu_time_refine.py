"""GPU bring-up helper: time the LSGAN refine generator NCSNpp(discriminative=True) at a given shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from universal_speech_enhancement_amd.hip_engine import HipScoreEngine
from universal_speech_enhancement_amd.testing import weights as tw
from universal_speech_enhancement_amd.testing import noise as tn

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
T = int(sys.argv[3]) if len(sys.argv) > 3 else 640
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 5
a = tw.REFINE
eng = HipScoreEngine(nf=a["nf"], ch_mult=a["ch_mult"], num_res_blocks=a["num_res_blocks"], precision=prec,
                     input_channels=2, conditional=False, scale_by_sigma=False)
eng.load_state_dict(tw.make_state_dict(4321, **a))
y = torch.from_numpy(tn.complex_normal(1, "y", (B, 1, 512, T))).cuda() * 0.5
out = eng.forward(y); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    out = eng.forward(y)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
fl = eng.flops_per_score()
print(f"[refine {prec}] B={B} T'={T}: {ms:.2f} ms / evaluation -> {fl / ms / 1e9:.1f} TFLOP/s ({fl / 1e12:.2f} TFLOP), "
      f"{B * T / ms * 1e3:.0f} padded frames/s, finite={torch.isfinite(torch.view_as_real(out)).all().item()}")
