#!/usr/bin/env python3
"""Golden-vector generator -- runs ONLY in the build container (needs /root/reference).

It imports the reference's own Python (with two in-memory stub modules for unrelated imports,
SURVEY.md section 8c), loads the deterministic weights of
``universal_speech_enhancement_amd.testing.weights`` into the reference modules, runs them on seeded
inputs and writes inputs + expected outputs to ``tests/golden/*.npz``.  Nothing of the reference's
source is copied: the fixtures hold data only.

    python oracle/gen_golden.py [--only NAME]

Fixtures
  fir.npz           upsample_2d / downsample_2d on [2,5,16,12]                      (G1)
  resblock_*.npz    ResnetBlockBigGANpp plain / widen / down / up / cat-input        (G1)
  resblock_grads_*  gradients of the plain / widen block from the reference's backward()  (8f4)
  attn.npz          AttnBlockpp [2,32,8,5]                                          (G1)
  forward_large.npz NCSNppLarge.forward [2,2,512,64], t in {1.0,0.5} and {0.03,0.2}  (G2)
  sampler_*.npz     get_pc_sampler with an analytic score_fn, N=7: reverse_diffusion x {none, langevin, ald},
                    euler_maruyama x {none, langevin}                                (G3)
  sample_e2e.npz    ScoreModel.sample, 0.4 s utterance, N=3, langevin x1             (G4)
  sample_cfg1.npz   BASELINE cfg1: 2 s utterance, N=5, reverse_diffusion+langevin    (G5, ~70 s)
  sample_denoised.npz ScoreModel.sample with condition="denoised", sde_input in {noisy, denoised}  (model_wrapper.py:283-328)
  forward_12m/6m.npz NCSNpp12M / NCSNpp6M (nf = 96) forward [2,2,512,64]                  
  lowprec_reference.npz  the reference's OWN 16-bit errors (CPU autocast forward bf16 / fp16, bf16-autocast backward) vs its fp32
                    results: the bounds the HIP 16-bit paths are held to
  refine.npz        LSGAN refine generator: NCSNpp(discriminative=True).forward [2,1,512,64] and
                    NCSNPP_Wrapper inference on 2 x 0.4 s                           
"""
import argparse
import os
import sys
import types
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
for _m in ("torchaudio", "pydub"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
sys.modules["pydub"].AudioSegment = object

import torch  # noqa: E402

from universal_speech_enhancement_amd.testing import noise as tnoise  # noqa: E402
from universal_speech_enhancement_amd.testing import weights as tw  # noqa: E402

from src.models.components.sgmse import sampling as ref_sampling  # noqa: E402
from src.models.components.sgmse.backbones.ncsnpp_utils import layerspp, up_or_down_sampling  # noqa: E402
from src.models.components.sgmse.model_wrapper import ScoreModel  # noqa: E402
from src.models.components.sgmse.sdes import OUVESDE  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
torch.set_num_threads(8)


def fill_module(mod: torch.nn.Module, seed: int, tag: str):
    """Deterministic small-module weights (stored in the fixture as well)."""
    sd = {}
    for k, v in mod.state_dict().items():
        n = v.numel()
        if v.ndim >= 2:
            bound = np.sqrt(3.0 / (sum(v.shape[:2]) / 2.0 * max(1, int(np.prod(v.shape[2:])))))
            a = (tw.uniform01(seed, tag + k, n) * 2 - 1) * bound
        elif "GroupNorm" in k and k.endswith("weight"):
            a = 1.0 + 0.2 * (tw.uniform01(seed, tag + k, n) - 0.5)
        else:
            a = 0.2 * (tw.uniform01(seed, tag + k, n) - 0.5)
        sd[k] = torch.from_numpy(a.astype(np.float32).reshape(tuple(v.shape)))
    mod.load_state_dict(sd)
    return {k: v.numpy() for k, v in sd.items()}


def rnd(seed, tag, shape, scale=1.0):
    return torch.from_numpy((tnoise.normal(seed, tag, int(np.prod(shape))) * scale).reshape(shape))


def gen_fir():
    x = rnd(1, "fir", (2, 5, 16, 12))
    up = up_or_down_sampling.upsample_2d(x, [1, 3, 3, 1], factor=2)
    dn = up_or_down_sampling.downsample_2d(x, [1, 3, 3, 1], factor=2)
    np.savez(os.path.join(OUT, "fir.npz"), x=x.numpy(), up=up.numpy(), down=dn.numpy())


def gen_resblocks():
    act = torch.nn.SiLU()
    cases = {"plain": dict(in_ch=16, out_ch=16), "widen": dict(in_ch=16, out_ch=32),
             "down": dict(in_ch=16, out_ch=16, down=True), "up": dict(in_ch=16, out_ch=16, up=True),
             "cat": dict(in_ch=48, out_ch=32)}
    for name, kw in cases.items():
        blk = layerspp.ResnetBlockBigGANpp(act=act, temb_dim=24, dropout=0.0, fir=True, fir_kernel=[1, 3, 3, 1],
                                          init_scale=0.0, skip_rescale=True, **kw).eval()
        w = fill_module(blk, 7, name)
        x = rnd(2, name + "x", (2, kw["in_ch"], 12, 10))
        temb = rnd(2, name + "t", (2, 24))
        with torch.no_grad():
            y = blk(x, temb)
        np.savez(os.path.join(OUT, f"resblock_{name}.npz"), x=x.numpy(), temb=temb.numpy(), y=y.numpy(),
                 **{"w." + k: v for k, v in w.items()})


def gen_resblock_grads():
    """Gradients of ResnetBlockBigGANpp from the reference's own backward() (the backward half of train_step), for the
    `plain` (residual), `widen` (1x1 shortcut), `down` and `up` (FIR-resampled) blocks of gen_resblocks - same module, weights and inputs; loss = sum(y * gy)."""
    act = torch.nn.SiLU()
    for name, kw in {"plain": dict(in_ch=16, out_ch=16), "widen": dict(in_ch=16, out_ch=32),
                     "down": dict(in_ch=16, out_ch=16, down=True), "up": dict(in_ch=16, out_ch=16, up=True)}.items():
        blk = layerspp.ResnetBlockBigGANpp(act=act, temb_dim=24, dropout=0.0, fir=True, fir_kernel=[1, 3, 3, 1],
                                          init_scale=0.0, skip_rescale=True, **kw).eval()
        fill_module(blk, 7, name)
        x = rnd(2, name + "x", (2, kw["in_ch"], 12, 10)).requires_grad_(True)
        temb = rnd(2, name + "t", (2, 24)).requires_grad_(True)
        y = blk(x, temb)
        gy = rnd(4, name + "gy", tuple(y.shape))
        (y * gy).sum().backward()
        np.savez(os.path.join(OUT, f"resblock_grads_{name}.npz"), gy=gy.numpy(), dx=x.grad.numpy(), dtemb=temb.grad.numpy(),
                 **{"d." + k: v.grad.numpy() for k, v in blk.named_parameters()})


def gen_attn_grads():
    """Gradients of AttnBlockpp (gen_attn's module, weights and input) from the reference's backward(); loss = sum(y * gy)."""
    blk = layerspp.AttnBlockpp(channels=32, skip_rescale=True, init_scale=0.0).eval()
    fill_module(blk, 9, "attn")
    x = rnd(3, "attnx", (2, 32, 8, 5)).requires_grad_(True)
    y = blk(x)
    gy = rnd(5, "attngy", tuple(y.shape))
    (y * gy).sum().backward()
    np.savez(os.path.join(OUT, "attn_grads.npz"), gy=gy.numpy(), dx=x.grad.numpy(), **{"d." + k: v.grad.numpy() for k, v in blk.named_parameters()})


def gen_attn():
    blk = layerspp.AttnBlockpp(channels=32, skip_rescale=True, init_scale=0.0).eval()
    w = fill_module(blk, 9, "attn")
    x = rnd(3, "attnx", (2, 32, 8, 5))
    with torch.no_grad():
        y = blk(x)
    np.savez(os.path.join(OUT, "attn.npz"), x=x.numpy(), y=y.numpy(), **{"w." + k: v for k, v in w.items()})


def build_reference_large(seed=1234):
    m = ScoreModel(backbone="ncsnpplarge", sde="ouve", t_eps=3e-2, condition="noisy", n_fft=1022, hop_length=160,
                   num_frames=512, window="hann", sde_input="noisy", predictor="reverse_diffusion",
                   corrector="langevin").eval()
    sd = tw.make_state_dict(seed, **tw.LARGE)
    m.score_net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    return m, tw.weights_checksum(sd)


def gen_forward_large(model=None):
    m, crc = model or build_reference_large()
    B, Fq, T = 2, 512, 64
    x = torch.from_numpy(tnoise.complex_normal(11, "fwd_x", (B, 2, Fq, T))) * 0.5
    outs = {}
    with torch.no_grad():
        for tag, tv in (("a", [1.0, 0.5]), ("b", [0.03, 0.2])):
            t = torch.tensor(tv, dtype=torch.float32)
            outs["t_" + tag] = t.numpy()
            outs["out_" + tag] = m.score_net(x, t).numpy()
    np.savez(os.path.join(OUT, "forward_large.npz"), x=x.numpy(), weights_seed=1234, weights_crc=crc, **outs)


class _Replay:
    """torch.randn_like replacement that replays a prepared list (and checks shapes)."""

    def __init__(self, draws):
        self.draws, self.i = draws, 0

    def __call__(self, like, **kw):
        z = torch.from_numpy(self.draws[self.i]); self.i += 1
        assert z.shape == like.shape and z.dtype == like.dtype
        return z


def gen_samplers():
    Y = torch.from_numpy(tnoise.complex_normal(21, "samp_y", (3, 1, 16, 8)))
    A = torch.from_numpy(tnoise.complex_normal(21, "samp_a", (1, 1, 16, 8)))

    def score_fn(x, t, score_conditioning=None, sde_input=None):  # cheap analytic stand-in
        return -(x - 0.8 * sde_input) / (0.1 + t[:, None, None, None] ** 2) + 0.05 * A * torch.tanh(x.abs())

    for corr in ("none", "langevin", "ald"):
        sde = OUVESDE(); sde.N = 7
        ndraw = 1 + 7 * (1 + (0 if corr == "none" else 2))
        draws = tnoise.sampler_noise(33, ndraw, tuple(Y.shape))
        orig = torch.randn_like
        torch.randn_like = _Replay(list(draws))
        try:
            x, nfe = ref_sampling.get_pc_sampler("reverse_diffusion", corr, sde=sde, score_fn=score_fn, y=Y, eps=0.03,
                                                 snr=0.5, corrector_steps=2, conditioning=[Y])()
        finally:
            torch.randn_like = orig
        np.savez(os.path.join(OUT, f"sampler_rd_{corr}.npz"), Y=Y.numpy(), A=A.numpy(), x=x.numpy(), nfe=nfe,
                 noise_seed=33, n_draws=ndraw, N=7, corrector_steps=2, snr=0.5, eps=0.03)


def gen_samplers_pf():
    """get_pc_sampler(..., probability_flow=True) through the reference's own sampler.  The flag reaches Predictor.__init__, which stores
    it and builds its reverse SDE as sde.reverse(score_fn) - without it (predictors.py:17) - so the outputs are those of the ordinary
    reverse SDE; the fixture pins exactly that (the ODE form of sdes.py:136-141, 165-172 is only reached by get_ode_sampler)."""
    Y = torch.from_numpy(tnoise.complex_normal(21, "samp_y", (3, 1, 16, 8)))
    A = torch.from_numpy(tnoise.complex_normal(21, "samp_a", (1, 1, 16, 8)))

    def score_fn(x, t, *args, score_conditioning=None, sde_input=None):
        cond = sde_input if sde_input is not None else (args[0] if args else (score_conditioning if score_conditioning is not None else Y))
        if isinstance(cond, (list, tuple)):
            cond = cond[0]
        return -(x - 0.8 * cond) / (0.1 + t[:, None, None, None] ** 2) + 0.05 * A * torch.tanh(x.abs())

    for pred in ("reverse_diffusion", "euler_maruyama"):
        sde = OUVESDE(); sde.N = 7
        ndraw = 1 + 7 * 3
        draws = tnoise.sampler_noise(35, ndraw, tuple(Y.shape))
        orig = torch.randn_like
        torch.randn_like = _Replay(list(draws))
        try:
            x, nfe = ref_sampling.get_pc_sampler(pred, "langevin", sde=sde, score_fn=score_fn, y=Y, eps=0.03, snr=0.5, corrector_steps=2,
                                                 probability_flow=True, conditioning=[Y])()
        finally:
            torch.randn_like = orig
        np.savez(os.path.join(OUT, f"sampler_pf_{pred}.npz"), Y=Y.numpy(), A=A.numpy(), x=x.numpy(), nfe=nfe,
                 noise_seed=35, n_draws=ndraw, N=7, corrector_steps=2, snr=0.5, eps=0.03)


def gen_samplers_em():
    """EulerMaruyamaPredictor (sampling/predictors.py:40-53 over RSDE.sde / rsde_parts, sdes.py:119-157).  On the reference's
    own predict path this predictor raises (rsde_parts calls score_model(x, t, conditioning) without sde_input, sdes.py:128);
    the sampler accepts any callable, so a score_fn with optional trailing arguments reaches the formula."""
    Y = torch.from_numpy(tnoise.complex_normal(21, "samp_y", (3, 1, 16, 8)))
    A = torch.from_numpy(tnoise.complex_normal(21, "samp_a", (1, 1, 16, 8)))

    def score_fn(x, t, score_conditioning=None, sde_input=None):
        c = score_conditioning[0]                            # == Y; sde_input is not passed on this route
        return -(x - 0.8 * c) / (0.1 + t[:, None, None, None] ** 2) + 0.05 * A * torch.tanh(x.abs())

    for corr in ("none", "langevin"):
        sde = OUVESDE(); sde.N = 7
        ndraw = 1 + 7 * (1 + (0 if corr == "none" else 2))
        draws = tnoise.sampler_noise(34, ndraw, tuple(Y.shape))
        orig = torch.randn_like
        torch.randn_like = _Replay(list(draws))
        try:
            x, nfe = ref_sampling.get_pc_sampler("euler_maruyama", corr, sde=sde, score_fn=score_fn, y=Y, eps=0.03,
                                                 snr=0.5, corrector_steps=2, conditioning=[Y])()
        finally:
            torch.randn_like = orig
        np.savez(os.path.join(OUT, f"sampler_em_{corr}.npz"), Y=Y.numpy(), A=A.numpy(), x=x.numpy(), nfe=nfe,
                 noise_seed=34, n_draws=ndraw, N=7, corrector_steps=2, snr=0.5, eps=0.03)


def _sample_case(m, crc, fname, n_utts, length, N, corrector_steps, seed):
    wav = torch.from_numpy(tnoise.synth_noisy_speech(n_utts, length, seed=seed))
    T = 1 + length // 160
    Tp = (T + 63) // 64 * 64
    ndraw = 1 + N * (1 + corrector_steps)
    draws = tnoise.sampler_noise(4321, ndraw, (n_utts, 1, 512, Tp))
    orig = torch.randn_like
    torch.randn_like = _Replay(list(draws))
    try:
        with torch.no_grad():
            out = m.sample({"perturbed": wav.clone()}, N=N, corrector_steps=corrector_steps, snr=0.5)["enhanced"]
    finally:
        torch.randn_like = orig
    noise_crc = f"{zlib.crc32(draws.tobytes()) & 0xFFFFFFFF:08x}"
    np.savez(os.path.join(OUT, fname), wav=wav.numpy(), enhanced=out.numpy(), N=N, corrector_steps=corrector_steps,
             snr=0.5, noise_seed=4321, n_draws=ndraw, noise_crc=noise_crc, weights_seed=1234, weights_crc=crc,
             predictor="reverse_diffusion", corrector="langevin")


def gen_sample_e2e(model=None):
    m, crc = model or build_reference_large()
    _sample_case(m, crc, "sample_e2e.npz", 1, 9600, 3, 1, seed=77)


def gen_sample_cfg1(model=None):
    m, crc = model or build_reference_large()
    _sample_case(m, crc, "sample_cfg1.npz", 1, 48000, 5, 1, seed=1234)


def gen_forward_small():
    """NCSNpp12M / NCSNpp6M (nf = 96; reference ncsnpp.py:527-559) forward at [2,2,512,64], t = (0.8, 0.1)."""
    from src.models.components.sgmse.backbones.ncsnpp import NCSNpp12M, NCSNpp6M
    x = torch.from_numpy(tnoise.complex_normal(17, "small_x", (2, 2, 512, 64))) * 0.5
    t = torch.tensor([0.8, 0.1])
    for name, cls, arch in (("12m", NCSNpp12M, tw.SMALL12M), ("6m", NCSNpp6M, tw.SMALL6M)):
        net = cls(input_channels=4).eval()
        sd = tw.make_state_dict(4242, **arch)
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        with torch.no_grad():
            out = net(x, t)
        # (the input is regenerated from its seed by the tests: tnoise.complex_normal(17, "small_x", (2, 2, 512, 64)) * 0.5)
        np.savez(os.path.join(OUT, f"forward_{name}.npz"), x_seed=17, t=t.numpy(), out=out.numpy(), weights_seed=4242,
                 weights_crc=tw.weights_checksum(sd))


def gen_sample_denoised(model=None):
    """ScoreModel.sample with condition="denoised" (the network is conditioned on the GAN-denoised spectrogram,
    model_wrapper.py:285-286) for both choices of sde_input (:293-300, 320-328): 0.4 s utterance, N=3, langevin x1."""
    _, crc = model or build_reference_large()
    wav = torch.from_numpy(tnoise.synth_noisy_speech(1, 9600, seed=77))
    fake = torch.from_numpy(tnoise.synth_noisy_speech(1, 9600, seed=78)) * 0.7 + 0.3 * wav      # stand-in for the GAN's output
    draws = tnoise.sampler_noise(4321, 1 + 3 * 2, (1, 1, 512, 64))
    out = {}
    for sde_input in ("noisy", "denoised"):
        torch.manual_seed(0)
        m = ScoreModel(backbone="ncsnpplarge", sde="ouve", t_eps=3e-2, condition="denoised", n_fft=1022, hop_length=160,
                       num_frames=512, window="hann", sde_input=sde_input, predictor="reverse_diffusion", corrector="langevin").eval()
        sd = tw.make_state_dict(1234, **tw.LARGE)
        m.score_net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        orig = torch.randn_like
        torch.randn_like = _Replay(list(draws))
        try:
            with torch.no_grad():
                b = m.sample({"perturbed": wav.clone(), "fake": fake.clone()}, N=3, corrector_steps=1, snr=0.5)
        finally:
            torch.randn_like = orig
        out[sde_input] = b["enhanced" if sde_input == "noisy" else "fake_sde_enhanced"].numpy()
    np.savez(os.path.join(OUT, "sample_denoised.npz"), wav=wav.numpy(), fake=fake.numpy(), out_sde_noisy=out["noisy"],
             out_sde_denoised=out["denoised"], N=3, corrector_steps=1, snr=0.5, noise_seed=4321, n_draws=7, weights_seed=1234,
             weights_crc=crc)


def gen_both(model=None):
    """condition="both" (the reference's constructor default, model_wrapper.py:43-46, 287-288): the 6-channel network
    NCSNppLarge(input_channels=6) sees cat[x, Y, Y_denoised].  One forward at [2,3,512,64] and ScoreModel.sample for both choices
    of sde_input (0.4 s utterance, N=3, langevin x1)."""
    sd = tw.make_state_dict(1234, **tw.LARGE_BOTH)
    crc = tw.weights_checksum(sd)
    x = torch.from_numpy(tnoise.complex_normal(17, "both_x", (2, 3, 512, 64))) * 0.5
    t = torch.tensor([0.7, 0.05], dtype=torch.float32)
    wav = torch.from_numpy(tnoise.synth_noisy_speech(1, 9600, seed=77))
    fake = torch.from_numpy(tnoise.synth_noisy_speech(1, 9600, seed=78)) * 0.7 + 0.3 * wav
    draws = tnoise.sampler_noise(4321, 1 + 3 * 2, (1, 1, 512, 64))
    out = {}
    for sde_input in ("noisy", "denoised"):
        torch.manual_seed(0)
        m = ScoreModel(backbone="ncsnpplarge", sde="ouve", t_eps=3e-2, condition="both", n_fft=1022, hop_length=160,
                       num_frames=512, window="hann", sde_input=sde_input, predictor="reverse_diffusion", corrector="langevin").eval()
        m.score_net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        if sde_input == "noisy":
            with torch.no_grad():
                fwd = m.score_net(x, t).numpy()
        orig = torch.randn_like
        torch.randn_like = _Replay(list(draws))
        try:
            with torch.no_grad():
                b = m.sample({"perturbed": wav.clone(), "fake": fake.clone()}, N=3, corrector_steps=1, snr=0.5)
        finally:
            torch.randn_like = orig
        out[sde_input] = b["enhanced" if sde_input == "noisy" else "fake_sde_enhanced"].numpy()
    np.savez(os.path.join(OUT, "both.npz"), x=x.numpy(), t=t.numpy(), fwd=fwd, wav=wav.numpy(), fake=fake.numpy(),
             out_sde_noisy=out["noisy"], out_sde_denoised=out["denoised"], N=3, corrector_steps=1, snr=0.5, noise_seed=4321, n_draws=7,
             weights_seed=1234, weights_crc=crc)


def gen_train_loss(model=None):
    """ScoreModel.train_step (model_wrapper.py:147-208) = the value validation_step / test_step log (SGMSE_module.py:56-63), with
    the three random draws pinned (np.random.uniform -> start, torch.rand -> t, torch.randn_like -> z).  num_frames=64 keeps the
    fixture small (target_len 10080 samples, T = 64 frames).  Case a: condition noisy, 12000-sample clips (random excerpt), mse;
    case b: condition both / sde_input denoised, 8000-sample clips (zero padding), mse and mae."""
    import src.models.components.sgmse.model_wrapper as MW
    out = {}
    for tag, cond, sde_in, L, arch in (("a", "noisy", "noisy", 12000, tw.LARGE), ("b", "both", "denoised", 8000, tw.LARGE_BOTH)):
        clean = torch.from_numpy(tnoise.synth_noisy_speech(2, L, seed=91))
        noisy = 0.8 * clean + 0.2 * torch.from_numpy(tnoise.synth_noisy_speech(2, L, seed=92))
        fake = 0.9 * clean + 0.1 * torch.from_numpy(tnoise.synth_noisy_speech(2, L, seed=93))
        t = torch.tensor([0.31, 0.87], dtype=torch.float32)
        z = torch.from_numpy(tnoise.complex_normal(55, "train_z_" + tag, (2, 1, 512, 64)))
        start = 1234
        sd = tw.make_state_dict(1234, **arch)
        for loss_type in (("mse",) if tag == "a" else ("mse", "mae")):
            torch.manual_seed(0)
            m = ScoreModel(backbone="ncsnpplarge", sde="ouve", t_eps=3e-2, condition=cond, loss_type=loss_type, n_fft=1022,
                           hop_length=160, num_frames=64, window="hann", sde_input=sde_in).eval()
            m.score_net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
            o_rand, o_randn, o_unif = torch.rand, torch.randn_like, MW.np.random.uniform
            torch.rand = lambda *a, **k: (t - m.t_eps) / (m.sde.T - m.t_eps)       # so that rand * (T - eps) + eps == t (to 1 ulp)
            torch.randn_like = lambda like, **k: z.clone()
            MW.np.random.uniform = lambda lo, hi: start
            try:
                with torch.no_grad():
                    loss = m.train_step({"clean": clean.clone(), "perturbed": noisy.clone(), "fake": fake.clone()})
            finally:
                torch.rand, torch.randn_like, MW.np.random.uniform = o_rand, o_randn, o_unif
            out[f"loss_{tag}_{loss_type}"] = np.float64(loss.item())
        out.update({f"clean_{tag}": clean.numpy(), f"noisy_{tag}": noisy.numpy(), f"fake_{tag}": fake.numpy(), f"t_{tag}": t.numpy(),
                    f"crc_{tag}": tw.weights_checksum(sd)})       # z: tnoise.complex_normal(55, "train_z_<tag>", (2, 1, 512, 64))
    np.savez(os.path.join(OUT, "train_loss.npz"), start=1234, num_frames=64, weights_seed=1234, z_seed=55, **out)


def gen_train_grads(model=None):
    """The gradient half of training_step (SGMSE_module.py:46-54): loss = ScoreModel.train_step(batch) exactly as in gen_train_loss
    (same clips, t, z, start, weights), then the reference's own loss.backward().  Stored per parameter: the L2 norm of the gradient
    ("n.<name>", float64); the full gradient of every tensor of <= 10000 elements ("g.<name>"); and, of the large ones, the leading
    [:4, :4, ...] corner ("c.<name>").  Case a: condition noisy / mse; case b: condition both / sde_input denoised / mae."""
    import src.models.components.sgmse.model_wrapper as MW
    for tag, cond, sde_in, L, arch, loss_type in (("a", "noisy", "noisy", 12000, tw.LARGE, "mse"),
                                                   ("b", "both", "denoised", 8000, tw.LARGE_BOTH, "mae")):
        clean = torch.from_numpy(tnoise.synth_noisy_speech(2, L, seed=91))
        noisy = 0.8 * clean + 0.2 * torch.from_numpy(tnoise.synth_noisy_speech(2, L, seed=92))
        fake = 0.9 * clean + 0.1 * torch.from_numpy(tnoise.synth_noisy_speech(2, L, seed=93))
        t = torch.tensor([0.31, 0.87], dtype=torch.float32)
        z = torch.from_numpy(tnoise.complex_normal(55, "train_z_" + tag, (2, 1, 512, 64)))
        start = 1234
        sd = tw.make_state_dict(1234, **arch)
        torch.manual_seed(0)
        m = ScoreModel(backbone="ncsnpplarge", sde="ouve", t_eps=3e-2, condition=cond, loss_type=loss_type, n_fft=1022,
                       hop_length=160, num_frames=64, window="hann", sde_input=sde_in).eval()
        m.score_net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        o_rand, o_randn, o_unif = torch.rand, torch.randn_like, MW.np.random.uniform
        torch.rand = lambda *a, **k: (t - m.t_eps) / (m.sde.T - m.t_eps)
        torch.randn_like = lambda like, **k: z.clone()
        MW.np.random.uniform = lambda lo, hi: start
        try:
            loss = m.train_step({"clean": clean.clone(), "perturbed": noisy.clone(), "fake": fake.clone()})
        finally:
            torch.rand, torch.randn_like, MW.np.random.uniform = o_rand, o_randn, o_unif
        loss.backward()
        out = {"loss": np.float64(loss.item())}
        for k, p in m.score_net.named_parameters():
            if p.grad is None:
                continue                                                   # the Fourier projection's W is not trained
            g = p.grad.detach()
            out["n." + k] = np.float64(g.double().norm().item())
            if g.numel() <= 10000:
                out["g." + k] = g.numpy().astype(np.float32)
            else:
                out["c." + k] = g[:4, :4].contiguous().numpy().astype(np.float32)
        np.savez(os.path.join(OUT, f"train_grads_{tag}.npz"), weights_seed=1234, z_seed=55, start=start, num_frames=64, t=t.numpy(),
                 crc=tw.weights_checksum(sd), **out)


def gen_refine(model=None):
    """LSGAN refine stage: NCSNPP_Wrapper(n_fft=1022, hop=160, num_frames=480) = NCSNpp(discriminative=True)
    between STFT glue (GAN/generator/ncsnpp/model_wrapper.py:19-121, configs/model/LSGAN.yaml:46-53)."""
    from src.models.components.GAN.generator.ncsnpp.model_wrapper import NCSNPP_Wrapper
    w = NCSNPP_Wrapper(n_fft=1022, hop_length=160, num_frames=480, window="hann", spec_factor=0.15, spec_abs_exponent=0.5).eval()
    sd = tw.make_state_dict(4321, **tw.REFINE)
    w.net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    crc = tw.weights_checksum(sd)
    x = torch.from_numpy(tnoise.complex_normal(13, "refine_x", (2, 1, 512, 64))) * 0.5
    wav = torch.from_numpy(tnoise.synth_noisy_speech(2, 9600, seed=77))
    with torch.no_grad():
        out = w.net(x)
        fake = w({"perturbed": wav.clone()})["fake"]
    np.savez(os.path.join(OUT, "refine.npz"), x=x.numpy(), out=out.numpy(), wav=wav.numpy(), fake=fake.numpy(),
             weights_seed=4321, weights_crc=crc)


def _lowprec_chain(m, out):
    """The chain figures of lowprec_reference.npz (see gen_lowprec); round 6 adds the 99.99th percentile of |error| / max|reference| -
    a statistic of the error's tail that, unlike the maximum over all elements, does not move by tens of per cent with the summation
    order of a GroupNorm partial sum (tests/lowprec.py)."""
    # the benchmarked sampler as a chain: 30 PC steps (reverse_diffusion + Langevin x1 = 60 evaluations, t down to 0.03, the score's
    # 1/t amplification included) on one 0.4 s utterance, the reference under bf16 / fp16 autocast vs its own fp32 run, same noise:
    # how far the REFERENCE drifts in 16 bits (spectrogram fed to spec_back and waveform; relative to the maximum and relative L2)
    N_chain = 30
    pooled = {(d, w): [] for d in ("bf16", "fp16") for w in ("spec", "wav")}
    # utterance 0 (seed 77 / noise 4321) is the one the round-4 maxima and L2 figures were generated on; three more utterances are pooled
    # with it for the percentile only (9 600 waveform samples hold 0.96 elements above their 99.99th percentile: one utterance's figure is
    # its second-largest error, as noisy as the maximum; each utterance's errors are taken relative to its own maximum)
    for u, (wseed, nseed) in enumerate(((77, 4321), (78, 4322), (79, 4323), (80, 4324))):
        wav = torch.from_numpy(tnoise.synth_noisy_speech(1, 9600, seed=wseed))
        draws = tnoise.sampler_noise(nseed, 1 + 2 * N_chain, (1, 1, 512, 64))
        chain = {}
        for dt_name, dt in (("fp32", None), ("bf16", torch.bfloat16), ("fp16", torch.float16)):
            grabbed = {}
            orig_back = m.spec_back
            m.spec_back = lambda spec, _o=orig_back, _g=grabbed: (_g.__setitem__("spec", spec.clone()), _o(spec))[1]
            orig = torch.randn_like
            torch.randn_like = _Replay(list(draws))
            try:
                with torch.no_grad():
                    if dt is None:
                        w = m.sample({"perturbed": wav.clone()}, N=N_chain, corrector_steps=1, snr=0.5)["enhanced"]
                    else:
                        with torch.autocast("cpu", dtype=dt):
                            w = m.sample({"perturbed": wav.clone()}, N=N_chain, corrector_steps=1, snr=0.5)["enhanced"]
            finally:
                torch.randn_like = orig
                m.spec_back = orig_back
            chain[dt_name] = (torch.view_as_real(grabbed["spec"].to(torch.complex64)).double(), w.double())
            print("  chain utterance", u, dt_name, "done", flush=True)
        for dt_name in ("bf16", "fp16"):
            for what, i in (("spec", 0), ("wav", 1)):
                ref, low = chain["fp32"][i], chain[dt_name][i]
                pooled[dt_name, what].append(((low - ref).abs() / ref.abs().max()).numpy().ravel())
                if u == 0:
                    out[f"chain_{dt_name}_{what}_relmax"] = np.float64((low - ref).abs().max() / ref.abs().max())
                    out[f"chain_{dt_name}_{what}_rell2"] = np.float64((low - ref).norm() / ref.norm())
    for (dt_name, what), errs in pooled.items():
        e = np.concatenate(errs)
        out[f"chain_{dt_name}_{what}_p9999"] = np.float64(np.quantile(e, 0.9999))
        out[f"chain_{dt_name}_{what}_p9999_per_utt"] = np.array([np.quantile(x, 0.9999) for x in errs])
        out[f"chain_{dt_name}_{what}_relmax_per_utt"] = np.array([x.max() for x in errs])
        print("  chain", dt_name, what, "p9999 pooled", float(out[f"chain_{dt_name}_{what}_p9999"]), "per utterance", out[f"chain_{dt_name}_{what}_p9999_per_utt"],
              "relmax per utterance", out[f"chain_{dt_name}_{what}_relmax_per_utt"], flush=True)
    out["chain_N"] = N_chain


def gen_lowprec_chain(model=None):
    """Re-run only the sampler-chain part of gen_lowprec and merge it into the existing lowprec_reference.npz: the figures that
    were there must reproduce (same reference, same inputs, same thread count), the percentile keys are added."""
    m, crc = model or build_reference_large()
    path = os.path.join(OUT, "lowprec_reference.npz")
    old = dict(np.load(path))
    assert str(old["weights_crc"]) == str(crc)
    out = {}
    _lowprec_chain(m, out)
    for k, v in out.items():
        if k in old and "_p9999" not in k and "_per_utt" not in k:
            assert abs(float(old[k]) - float(v)) <= 0.02 * abs(float(old[k])) + 1e-12, (k, float(old[k]), float(v))
    kept = {k: v for k, v in old.items()}
    kept.update({k: v for k, v in out.items() if "_p9999" in k or "_per_utt" in k})      # the stored maxima / L2 figures stay as generated in round 4
    np.savez(path, **kept)
    print("  merged:", {k: (float(v) if np.ndim(v) == 0 else np.asarray(v).tolist()) for k, v in kept.items() if k.startswith("chain_")})


def gen_lowprec(model=None):
    """What the REFERENCE itself does in 16-bit arithmetic - the yardstick for the 16-bit tolerances of the HIP path (tests assert
    "no worse than the reference's own reduced-precision run", never "2x what we measured last time").
      forward: NCSNppLarge.forward on the forward_large inputs under torch.autocast("cpu", bfloat16 / float16) against its own fp32
               output: error relative to the maximum and relative L2, per t-pair.
      training: the reference's loss.backward() of gen_train_grads case a under bf16 autocast against its fp32 gradients: loss,
               worst gradient-norm error, and per-tensor relative L2 errors (all 616 tensors: max, 99th / 95th percentile, median)."""
    m, crc = model or build_reference_large()
    g = dict(np.load(os.path.join(OUT, "forward_large.npz")))
    x = torch.from_numpy(g["x"])
    out = {}
    for dt_name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
        for tag in ("a", "b"):
            t = torch.from_numpy(g["t_" + tag])
            ref = torch.from_numpy(g["out_" + tag])
            try:
                with torch.no_grad(), torch.autocast("cpu", dtype=dt):
                    low = m.score_net(x, t)
            except Exception as e:   # noqa: BLE001
                print("  autocast", dt_name, "unsupported on this CPU build:", e)
                continue
            d = (torch.view_as_real(low.to(torch.complex64)) - torch.view_as_real(ref)).double()
            out[f"fwd_{dt_name}_relmax_{tag}"] = np.float64(d.abs().max() / ref.abs().max())
            out[f"fwd_{dt_name}_rell2_{tag}"] = np.float64(d.norm() / torch.view_as_real(ref).double().norm())
            print(f"  forward {dt_name} {tag}: relmax {out[f'fwd_{dt_name}_relmax_{tag}']:.4g} rel-L2 {out[f'fwd_{dt_name}_rell2_{tag}']:.4g}", flush=True)
    _lowprec_chain(m, out)
    # the refine generator (NCSNpp(discriminative=True), LSGAN stage) under bf16 autocast: the refine.npz input and the T' = 128 input of
    # tests/test_hip_parity.py::test_refine_generator_long_sequence_attention (1024-token attention)
    sdr = tw.make_state_dict(4321, **tw.REFINE)
    try:
        gr = dict(np.load(os.path.join(OUT, "refine.npz")))
        import src.models.components.GAN.generator.ncsnpp.model_wrapper as GW
        wr = GW.NCSNPP_Wrapper(n_fft=1022, hop_length=160, num_frames=480, window="hann", spec_factor=0.15, spec_abs_exponent=0.5).eval()
        wr.net.load_state_dict({k: torch.from_numpy(v) for k, v in sdr.items()}, strict=True)
        for tag, xin in (("golden", torch.from_numpy(gr["x"])), ("t128", torch.from_numpy(tnoise.complex_normal(31, "Y", (1, 1, 512, 128))) * 0.5)):
            with torch.no_grad():
                r32 = wr.net(xin)
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    r16 = wr.net(xin)
            d = (torch.view_as_real(r16.to(torch.complex64)) - torch.view_as_real(r32)).double()
            out[f"refine_bf16_relmax_{tag}"] = np.float64(d.abs().max() / r32.abs().max())
            print(f"  refine bf16 {tag}: relmax {out[f'refine_bf16_relmax_{tag}']:.4g}", flush=True)
    except Exception as e:   # noqa: BLE001
        print("  refine figures skipped:", repr(e))
    # training gradients under bf16 autocast (case a of gen_train_grads)
    import src.models.components.sgmse.model_wrapper as MW
    L, arch = 12000, tw.LARGE
    clean = torch.from_numpy(tnoise.synth_noisy_speech(2, L, seed=91))
    noisy = 0.8 * clean + 0.2 * torch.from_numpy(tnoise.synth_noisy_speech(2, L, seed=92))
    fake = 0.9 * clean + 0.1 * torch.from_numpy(tnoise.synth_noisy_speech(2, L, seed=93))
    t = torch.tensor([0.31, 0.87], dtype=torch.float32)
    z = torch.from_numpy(tnoise.complex_normal(55, "train_z_a", (2, 1, 512, 64)))
    start = 1234
    sd = tw.make_state_dict(1234, **arch)
    grads = {}
    for mode in ("fp32", "bf16"):
        torch.manual_seed(0)
        mm = ScoreModel(backbone="ncsnpplarge", sde="ouve", t_eps=3e-2, condition="noisy", loss_type="mse", n_fft=1022,
                        hop_length=160, num_frames=64, window="hann", sde_input="noisy").eval()
        mm.score_net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
        o_rand, o_randn, o_unif = torch.rand, torch.randn_like, MW.np.random.uniform
        torch.rand = lambda *a, **k: (t - mm.t_eps) / (mm.sde.T - mm.t_eps)
        torch.randn_like = lambda like, **k: z.clone()
        MW.np.random.uniform = lambda lo, hi: start
        try:
            if mode == "bf16":
                with torch.autocast("cpu", dtype=torch.bfloat16):
                    loss = mm.train_step({"clean": clean.clone(), "perturbed": noisy.clone(), "fake": fake.clone()})
            else:
                loss = mm.train_step({"clean": clean.clone(), "perturbed": noisy.clone(), "fake": fake.clone()})
        finally:
            torch.rand, torch.randn_like, MW.np.random.uniform = o_rand, o_randn, o_unif
        loss.backward()
        grads[mode] = (float(loss.item()), {k: p.grad.detach().double() for k, p in mm.score_net.named_parameters() if p.grad is not None})
    (l32, g32), (l16, g16) = grads["fp32"], grads["bf16"]
    rel = []
    worst_n = 0.0
    for k, a in g32.items():
        na = float(a.norm())
        if na < 1e-12 or k.endswith("NIN_1.b"):      # the key bias of an attention block: its gradient is analytically zero (softmax is
            continue                                  # shift-invariant), what is stored is rounding noise
        rel.append(float((g16[k] - a).norm()) / na)
        worst_n = max(worst_n, abs(float(g16[k].norm()) - na) / na)
    rel = np.sort(np.array(rel))
    out.update(train_bf16_loss_rel=np.float64(abs(l16 - l32) / abs(l32)), train_bf16_norm_rel_max=np.float64(worst_n),
               train_bf16_tensor_rell2_max=np.float64(rel[-1]), train_bf16_tensor_rell2_p99=np.float64(rel[int(0.99 * (len(rel) - 1))]),
               train_bf16_tensor_rell2_p95=np.float64(rel[int(0.95 * (len(rel) - 1))]), train_bf16_tensor_rell2_median=np.float64(np.median(rel)),
               train_n_tensors=len(rel))
    print("  training bf16 autocast:", {k: float(v) for k, v in out.items() if k.startswith("train_")}, flush=True)
    np.savez(os.path.join(OUT, "lowprec_reference.npz"), weights_seed=1234, weights_crc=crc, **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    small = {"fir": gen_fir, "resblocks": gen_resblocks, "resblock_grads": gen_resblock_grads, "attn": gen_attn, "attn_grads": gen_attn_grads, "samplers": gen_samplers, "samplers_em": gen_samplers_em, "samplers_pf": gen_samplers_pf,
             "refine": gen_refine, "forward_small": gen_forward_small, "both": gen_both,
             "train_loss": gen_train_loss, "train_grads": gen_train_grads}
    big = {"forward_large": gen_forward_large, "sample_e2e": gen_sample_e2e, "sample_cfg1": gen_sample_cfg1,
           "sample_denoised": gen_sample_denoised, "lowprec": gen_lowprec, "lowprec_chain": gen_lowprec_chain}
    todo = [a.only] if a.only else list(small) + [n for n in big if n != "lowprec_chain"]
    model = build_reference_large() if any(n in big for n in todo) else None
    for n in todo:
        print("generating", n, flush=True)
        (small[n]() if n in small else big[n](model))
    print("done")
