"""GPU bring-up helper: per-tap comparison of the HIP forward against the CPU oracle (not a test)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from universal_speech_enhancement_amd.hip_engine import HipScoreEngine
from universal_speech_enhancement_amd.testing import weights as tw
from oracle import ncsnpp_oracle as no

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
g = dict(np.load(os.path.join(ROOT, "tests/golden/forward_large.npz")))
sdn = tw.make_state_dict(1234, **tw.LARGE)
assert tw.weights_checksum(sdn) == str(g["weights_crc"])
eng = HipScoreEngine(precision=prec)
t0 = time.time(); eng.load_state_dict(sdn); print("weights uploaded in %.1fs" % (time.time() - t0))
x = torch.from_numpy(g["x"]).cuda()
sd = no.to_torch(sdn)
for tag in ("a", "b"):
    t = torch.from_numpy(g["t_" + tag])
    out = eng.score(x[:, 0:1].contiguous(), x[:, 1:2].contiguous(), t.cuda())
    torch.cuda.synchronize()
    ref = -torch.from_numpy(g["out_" + tag])   # engine returns the score = -net
    err = (out.cpu() - ref).abs().max().item() / ref.abs().max().item()
    print(f"[{prec}] t={t.tolist()} score rel-max err {err:.3e}  (|ref|max {ref.abs().max():.3f})")
    taps = {}
    with torch.no_grad():
        no.ncsnpp_forward(sd, torch.from_numpy(g["x"]), t, taps=taps)
    for name in ("h_in", "pre_attn", "post_attn", "pyramid"):
        d = eng.debug_tensor(name).cpu().permute(0, 3, 1, 2)
        r = taps[name]
        e = (d - r).abs().max().item() / r.abs().max().item()
        print(f"    tap {name:10s} shape {tuple(d.shape)} rel-max err {e:.3e}")
print("flops/score %.3f GF, workspace %.1f MB" % (eng.flops_per_score() / 1e9, eng.workspace_bytes() / 1e6))
