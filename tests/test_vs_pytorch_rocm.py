"""The same network in plain PyTorch-ROCm (the oracle's functional restatement of the reference modules, run ON THE GPU: MIOpen convolutions,
ATen GroupNorm / SiLU / softmax - what the reference itself would execute on this machine) timed against the HIP path.  Prints both
([measured] lines) and asserts the HIP path is the faster one.  Default: a small shape (seconds).  USE_VS_TORCH_FULL=1: the benchmark
shapes - one score evaluation of 8 x 640 frames and one training step (forward + backward) of 4 x 512 frames; the numbers quoted in
DESIGN.md / profiles/r3_vs_pytorch_rocm.txt come from that mode."""
import os
import time

import pytest
import torch

import lowprec as lp
from oracle import ncsnpp_oracle as no
from universal_speech_enhancement_amd.testing import noise as tnoise
from universal_speech_enhancement_amd.testing import weights as tw

pytestmark = pytest.mark.gpu
FULL = os.environ.get("USE_VS_TORCH_FULL") == "1"


def _timed(fn, n):
    fn(); torch.cuda.synchronize()                                   # warm-up (MIOpen picks / compiles its kernels here)
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def test_score_evaluation_and_training_step_against_pytorch_rocm_eager(monkeypatch):
    from universal_speech_enhancement_amd.sgmse.backbones import BackboneRegistry
    fk = no._fir_kernel
    monkeypatch.setattr(no, "_fir_kernel", lambda *a, **k: fk(*a, **k).cuda())      # the oracle builds its FIR taps on the CPU
    sd_np = tw.make_state_dict(1234, **tw.LARGE)
    sd = {k: v.cuda() for k, v in no.to_torch(sd_np).items()}
    B, T = (8, 640) if FULL else (2, 64)
    x = (torch.from_numpy(tnoise.complex_normal(3, "vs_x", (B, 2, 512, T))) * 0.5).cuda()
    t = torch.full((B,), 0.5, device="cuda")
    net = BackboneRegistry.get_by_name("ncsnpplarge")(input_channels=4, precision="bf16")
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    net = net.cuda()
    with torch.no_grad():
        ref = no.ncsnpp_forward(sd, x, t)
        t_torch = _timed(lambda: no.ncsnpp_forward(sd, x, t), 3)
        with torch.autocast("cuda", dtype=torch.float16):
            t_torch16 = _timed(lambda: no.ncsnpp_forward(sd, x, t), 3)
        got = net(x, t)
        t_hip = _timed(lambda: net(x, t), 10)
        net.precision = "fp32"; net._engine = None
        t_hip32 = _timed(lambda: net(x, t), 3)
        got32 = net(x, t)
    assert float((got32 - ref).abs().max()) < 5e-4 * float(ref.abs().max())
    assert float((got - ref).abs().max()) < lp.fwd_bound("bf16", 1.25) * float(ref.abs().max())      # tests/lowprec.py: 1.25 x the reference's own bf16 autocast error (other inputs than the fixture's)
    print(f"\n[measured] one score evaluation, B={B} x {T} frames: PyTorch-ROCm eager fp32 {t_torch * 1e3:.1f} ms, fp16 autocast {t_torch16 * 1e3:.1f} ms; "
          f"HIP path fp32 {t_hip32 * 1e3:.1f} ms, bf16 {t_hip * 1e3:.1f} ms ({t_torch / t_hip:.1f}x the fp32 eager run)")
    if FULL:      # wall-clock comparisons are printed always, asserted only on request: the driver runs pytest -x on a shared box
        assert t_hip < min(t_torch, t_torch16) and t_hip32 < t_torch
    del got, got32, ref
    # ---- one training step (forward + backward of a squared-error loss on the network output; no optimiser) -------------------------
    Bt, Tt = (4, 512) if FULL else (2, 64)
    xt = (torch.from_numpy(tnoise.complex_normal(4, "vs_xt", (Bt, 2, 512, Tt))) * 0.5).cuda()
    tt = torch.full((Bt,), 0.5, device="cuda")
    ref_p = {k: (v.clone().requires_grad_(k != "all_modules.0.W")) for k, v in sd.items()}

    def step_torch():
        for p in ref_p.values():
            p.grad = None
        no.ncsnpp_forward(ref_p, xt, tt).abs().square().mean().backward()
    net.requires_grad_(True)

    def step_hip():
        net.zero_grad(set_to_none=True)
        net(xt, tt).abs().square().mean().backward()
    t_ts = _timed(step_torch, 2)
    net.train_precision = "fp32"
    t_hs32 = _timed(step_hip, 3)
    g_ref = ref_p["all_modules.5.Conv_0.weight"].grad
    g_hip = dict(net.named_parameters())["all_modules.5.Conv_0.weight"].grad
    assert float((g_hip - g_ref).abs().max()) < 1e-3 * float(g_ref.abs().max())
    net.train_precision = "bf16"
    t_hs16 = _timed(step_hip, 3)
    print(f"[measured] one training step (forward + backward), B={Bt} x {Tt} frames: PyTorch-ROCm eager fp32 {t_ts * 1e3:.0f} ms; "
          f"HIP path fp32 {t_hs32 * 1e3:.0f} ms, bf16 mixed {t_hs16 * 1e3:.0f} ms")
    if FULL:
        assert t_hs16 < t_ts and t_hs32 < t_ts * 1.5
