"""How much would cross-evaluation pipelining of the sub-batch streams buy?  Three independent engines (3 + 3 + 2 items, no internal split) on three
streams, N score evaluations each with NO join between evaluations, against the library's own 3-stream evaluation of the 8 items (fork / join
per evaluation).  usage: python scripts/pipeline_probe.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from universal_speech_enhancement_amd.hip_engine import HipScoreEngine, set_option
from universal_speech_enhancement_amd.testing import weights as tw, noise as tn
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
sd = tw.make_state_dict(1234, **tw.LARGE)
T = 640
def mk(B, split):
    set_option("subbatch", split)
    e = HipScoreEngine(precision="bf16"); e.load_state_dict(sd); e.plan(B, T)
    x = torch.from_numpy(tn.complex_normal(1, f"x{B}", (B, 1, 512, T))).cuda() * 0.5
    y = torch.from_numpy(tn.complex_normal(1, f"y{B}", (B, 1, 512, T))).cuda() * 0.5
    t = torch.full((B,), 0.5).cuda()
    e.score(x, y, t); torch.cuda.synchronize()
    return e, x, y, t
whole = mk(8, -1)
parts = [mk(b, 0) for b in (3, 3, 2)]
streams = [torch.cuda.Stream() for _ in parts]
def run_whole(n):
    e, x, y, t = whole
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): e.score(x, y, t)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def run_parts(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        for (e, x, y, t), s in zip(parts, streams):
            with torch.cuda.stream(s): e.score(x, y, t)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(2):
    print(f"library, 8 items as 3 + 3 + 2 with fork / join per evaluation: {run_whole(N):.2f} ms per evaluation of 8 items")
    print(f"three free-running streams (3, 3, 2 items), no joins:          {run_parts(N):.2f} ms per 8 items")
