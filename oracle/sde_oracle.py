"""ORACLE (test infrastructure, not product code) -- CPU restatement of the OUVE SDE, the
predictor / corrector update rules, the PC sampling loop and the STFT glue around it.

Imported only by ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.
Pinned against the reference by ``oracle/gen_golden.py`` -> ``tests/golden/sampler_*.npz`` and
``sample_e2e.npz`` (the reference has no tests of its own for this path).

Reference paths are relative to ``/root/reference/src/models/components/sgmse/``.

Noise handling: every Gaussian draw goes through ``NoiseSource`` so that a run can either consume a
``torch.Generator`` (and record what it drew) or replay a recorded list -- the order of draws is the
reference's: prior z0 (sdes.py:254), then per step corrector draws (correctors.py:54) followed by the
predictor draw (predictors.py:63).
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

THETA, SIGMA_MIN, SIGMA_MAX = 1.5, 0.05, 0.5  # sdes.py:184
LOGSIG = np.log(SIGMA_MAX / SIGMA_MIN)         # sdes.py:205


class NoiseSource:
    """complex64 standard-normal draws shaped like ``like`` -- ``torch.randn_like(complex)`` semantics
    (variance 1/2 per component)."""

    def __init__(self, generator: Optional[torch.Generator] = None, replay: Optional[Sequence[torch.Tensor]] = None):
        self.gen = generator
        self.replay = list(replay) if replay is not None else None
        self.drawn: List[torch.Tensor] = []
        self._i = 0

    def __call__(self, like: torch.Tensor) -> torch.Tensor:
        if self.replay is not None:
            z = self.replay[self._i]
            self._i += 1
        else:
            # torch.randn_like(complex) == view_as_complex(randn(..., 2)) * sqrt(0.5) with identical
            # generator consumption (SURVEY.md appendix A probe).
            r = torch.randn(*like.shape, 2, generator=self.gen, dtype=torch.float32)
            z = torch.view_as_complex(r) * math.sqrt(0.5)
        self.drawn.append(z)
        return z


def ouve_std(t: torch.Tensor) -> torch.Tensor:
    """OUVESDE._std (sdes.py:231-243)."""
    return torch.sqrt(
        (SIGMA_MIN ** 2 * torch.exp(-2 * THETA * t) * (torch.exp(2 * (THETA + LOGSIG) * t) - 1) * LOGSIG)
        / (THETA + LOGSIG)
    )


def ouve_sde(x, t, y):
    """OUVESDE.sde (sdes.py:216-224): drift theta (y - x), diffusion sigma(t) sqrt(2 logsig)."""
    drift = THETA * (y - x)
    sigma = SIGMA_MIN * (SIGMA_MAX / SIGMA_MIN) ** t
    return drift, sigma * np.sqrt(2 * LOGSIG)


def _bc(v: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    return v.view(*v.size(), *((1,) * (x.ndim - v.ndim))) if v.ndim < x.ndim else v


def predictor_reverse_diffusion(x, t, y, score, N, noise: NoiseSource):
    """ReverseDiffusionPredictor.update_fn (sampling/predictors.py:61-68) with RSDE.discretize
    (sdes.py:159-173) over SDE.discretize (sdes.py:88-92)."""
    dt = 1 / N
    drift, diffusion = ouve_sde(x, t, y)
    f = drift * dt
    G = _bc(diffusion * torch.sqrt(torch.tensor(dt)), x)
    rev_f = f - G ** 2 * score(x, t)
    z = noise(x)
    x_mean = x - rev_f
    return x_mean + G * z, x_mean


def predictor_euler_maruyama(x, t, y, score, N, noise: NoiseSource):
    """EulerMaruyamaPredictor.update_fn (sampling/predictors.py:44-53) with RSDE.sde / rsde_parts
    (sdes.py:119-157).  NB: on the reference's conditioned predict path this predictor raises a
    TypeError (sdes.py:128 omits sde_input); the formula is what the code would compute."""
    dt = -1.0 / N
    z = noise(x)
    drift, diffusion = ouve_sde(x, t, y)
    g = _bc(diffusion, x)
    total = drift - g ** 2 * score(x, t)
    x_mean = x + total * dt
    return x_mean + g * np.sqrt(-dt) * z, x_mean


def corrector_langevin(x, t, y, score, snr, n_steps, noise: NoiseSource):
    """LangevinCorrector.update_fn (sampling/correctors.py:45-63): step size from BATCH-MEAN norms."""
    x_mean = x
    for _ in range(n_steps):
        grad = score(x, t)
        z = noise(x)
        grad_norm = torch.norm(grad.reshape(grad.shape[0], -1), dim=-1).mean()
        noise_norm = torch.norm(z.reshape(z.shape[0], -1), dim=-1).mean()
        step = _bc(((snr * noise_norm / grad_norm) ** 2 * 2).unsqueeze(0), x)
        x_mean = x + step * grad
        x = x_mean + z * torch.sqrt(step * 2)
    return x, x_mean


def corrector_ald(x, t, y, score, snr, n_steps, noise: NoiseSource):
    """AnnealedLangevinDynamics.update_fn (sampling/correctors.py:79-98): step = 2 (snr std(t))^2."""
    std = ouve_std(t)
    x_mean = x
    for _ in range(n_steps):
        grad = score(x, t)
        z = noise(x)
        step = _bc((snr * std) ** 2 * 2, x)
        x_mean = x + step * grad
        x = x_mean + z * torch.sqrt(step * 2)
    return x, x_mean


PREDICTORS = {"reverse_diffusion": predictor_reverse_diffusion, "euler_maruyama": predictor_euler_maruyama,
              "none": lambda x, t, y, score, N, noise: (x, x)}
CORRECTORS = {"langevin": corrector_langevin, "ald": corrector_ald,
              "none": lambda x, t, y, score, snr, n_steps, noise: (x, x)}


def pc_sampler(score_fn: Callable, y: torch.Tensor, N: int, predictor="reverse_diffusion", corrector="none",
               corrector_steps=1, snr=0.5, eps=3e-2, noise: Optional[NoiseSource] = None, denoise=True):
    """sampling/__init__.py:59-71 (pc_sampler closure) with OUVESDE.prior_sampling (sdes.py:248-254).
    ``score_fn(x, t) -> score`` already closes over the conditioning.  Returns (x_result, nfe)."""
    noise = noise or NoiseSource(torch.Generator().manual_seed(0))
    B = y.shape[0]
    std1 = ouve_std(torch.ones(B))
    xt = y + noise(y) * std1[:, None, None, None]
    timesteps = torch.linspace(1, eps, N)
    xt_mean = xt
    n_corr = 0 if corrector == "none" else corrector_steps  # NoneCorrector sets n_steps=0 (correctors.py:106)
    with torch.no_grad():
        for i in range(N):
            vec_t = torch.ones(B) * timesteps[i]
            xt, xt_mean = CORRECTORS[corrector](xt, vec_t, y, score_fn, snr, n_corr, noise)
            xt, xt_mean = PREDICTORS[predictor](xt, vec_t, y, score_fn, N, noise)
    return (xt_mean if (denoise and N) else xt), N * (n_corr + 1)


# ------------------------------------------------------------------------------------------------
# STFT glue (model_wrapper.py:92-122, 262-329; util/other.py:128-135)
# ------------------------------------------------------------------------------------------------

def spec_fwd(spec, factor=0.15, e=0.5):
    """ScoreModel.spec_fwd (model_wrapper.py:92-96)."""
    if e != 1:
        spec = spec.abs() ** e * torch.exp(1j * spec.angle())
    return spec * factor


def spec_back(spec, factor=0.15, e=0.5):
    """ScoreModel.spec_back (model_wrapper.py:98-103)."""
    spec = spec / factor
    if e != 1:
        spec = spec.abs() ** (1 / e) * torch.exp(1j * spec.angle())
    return spec


def pad_spec(Y):
    """util/other.py:128-135: zero-pad the frame axis on the right to a multiple of 64."""
    T = Y.size(3)
    pad = (64 - T % 64) % 64
    return torch.nn.functional.pad(Y, (0, pad, 0, 0))


def stft(sig, n_fft=1022, hop=160):
    """ScoreModel.stft (model_wrapper.py:116-118): periodic hann, center=True."""
    return torch.stft(sig, n_fft=n_fft, hop_length=hop, window=torch.hann_window(n_fft, periodic=True),
                      center=True, return_complex=True)


def istft(spec, length, n_fft=1022, hop=160):
    """ScoreModel.istft (model_wrapper.py:120-122)."""
    return torch.istft(spec, n_fft=n_fft, hop_length=hop, window=torch.hann_window(n_fft, periodic=True),
                       center=True, length=length)


def score_model_sample(net_forward: Callable, wav: torch.Tensor, N=50, predictor="reverse_diffusion",
                       corrector="none", corrector_steps=1, snr=0.5, t_eps=3e-2,
                       noise: Optional[NoiseSource] = None):
    """ScoreModel.sample for condition='noisy', sde_input='noisy' (model_wrapper.py:262-329) with
    forward_score = -score_net(cat([x, Y]), t) (model_wrapper.py:135-141).
    ``net_forward(x_c64[B,2,F,T'], t[B]) -> c64[B,1,F,T']`` is the backbone."""
    T_orig = wav.size(1)
    Y = pad_spec(spec_fwd(stft(wav)).unsqueeze(1))

    def score(x, t):
        return -net_forward(torch.cat([x, Y], dim=1), t)

    sample, nfe = pc_sampler(score, Y, N, predictor, corrector, corrector_steps, snr, t_eps, noise)
    return istft(spec_back(sample.squeeze(1)), T_orig), sample, Y, nfe


def score_model_train_loss(net_forward: Callable, clean, perturbed, t, z, start, fake=None, condition="noisy", sde_input="noisy",
                           num_frames=64, hop=160, loss_type="mse", theta=1.5):
    """ScoreModel.train_step (model_wrapper.py:147-208) + _loss (:124-133) with the random draws (t [B], complex z, crop start)
    given: crop / pad to target_len = (num_frames - 1) hop, spectrograms, x_t = mean + std z with OUVESDE.marginal_prob
    (sdes.py:226-246), err = score(x_t, t) std + z.  ``net_forward(x_c64[B,2 or 3,F,T], t[B])`` is the backbone."""
    target_len = (num_frames - 1) * hop
    cur = clean.size(-1)
    pad = max(target_len - cur, 0)
    if pad == 0:
        cut = lambda a: a[..., start:start + target_len]                                  # noqa: E731
    else:
        cut = lambda a: torch.nn.functional.pad(a, (pad // 2, pad // 2 + (pad % 2)))       # noqa: E731
    X = spec_fwd(stft(cut(clean))).unsqueeze(1)
    Y = spec_fwd(stft(cut(perturbed))).unsqueeze(1)
    Yd = None if fake is None else spec_fwd(stft(cut(fake))).unsqueeze(1)
    ysde = Yd if sde_input == "denoised" else Y
    w = torch.exp(-theta * t)[:, None, None, None]
    mean = w * X + (1 - w) * ysde                                                          # OUVESDE._mean (sdes.py:226-229)
    sig = ouve_std(t)[:, None, None, None]
    xt = mean + sig * z
    cond = {"noisy": [Y], "denoised": [Yd], "both": [Y, Yd]}[condition]
    score = -net_forward(torch.cat([xt] + cond, dim=1), t)
    err = score * sig + z
    losses = err.abs() ** 2 if loss_type == "mse" else err.abs()
    return torch.mean(0.5 * torch.sum(losses.reshape(losses.shape[0], -1), dim=-1))


def refine_generator(net_forward: Callable, wav: torch.Tensor, n_fft=1022, hop=160):
    """NCSNPP_Wrapper.forward, inference branch (GAN/generator/ncsnpp/model_wrapper.py:114-121): the LSGAN refine stage
    that follows the sampler in the reference's documented pipeline (README.md:175-178; LSGAN_module.py:139-155).
    ``net_forward(Y_c64[B,1,F,T']) -> c64[B,1,F,T']`` is NCSNpp(discriminative=True)."""
    T_orig = wav.size(1)
    Y = pad_spec(spec_fwd(stft(wav, n_fft, hop)).unsqueeze(1))
    out = net_forward(Y)
    return istft(spec_back(out.squeeze(1)), T_orig, n_fft, hop), out, Y

