"""Size of the packed device weight blob per precision (the one RCCL broadcast of a multi-GPU job)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from universal_speech_enhancement_amd.hip_engine import HipScoreEngine
from universal_speech_enhancement_amd.testing import weights
sd = weights.make_state_dict(1234)
for prec in ("bf16", "fp32"):
    e = HipScoreEngine(precision=prec)
    e.load_state_dict(sd)
    n = e.weight_blob().numel()
    print(prec, n, "bytes =", round(n / 1e9, 3), "GB")
    e.close()
