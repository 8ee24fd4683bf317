"""Portable seeded Gaussian noise and synthetic 24 kHz 'noisy speech' for parity runs and benchmarks.

The CPU ``torch.Generator`` stream cannot be reproduced on the device (and is not guaranteed identical
across CPU vendors), so parity runs inject noise produced here: counter-hash uniforms (exact, see
``weights.uniform01``) through a float64 Box-Muller, rounded to float32.  The same call gives the same
tensor (to within a final-rounding ulp) in the build container and on the GPU box, so golden vectors
only need to store seeds, not the 21 MB-per-draw noise tensors.
"""
from __future__ import annotations

import numpy as np

from .weights import uniform01


def normal(seed: int, stream: str, n: int) -> np.ndarray:
    """n float32 N(0,1) samples."""
    u1 = uniform01(seed, stream, n, lane=101)
    u2 = uniform01(seed, stream, n, lane=102)
    r = np.sqrt(-2.0 * np.log(1.0 - u1))  # 1-u1 in (0, 1]
    return (r * np.cos(2.0 * np.pi * u2)).astype(np.float32)


def complex_normal(seed: int, stream: str, shape) -> np.ndarray:
    """complex64 array, E|z|^2 = 1 (variance 1/2 per component) == torch.randn_like(complex) law
    (reference sdes.py:254, predictors.py:63, correctors.py:54)."""
    n = int(np.prod(shape))
    g = normal(seed, stream, 2 * n).reshape(n, 2) * np.float32(np.sqrt(0.5))
    return np.ascontiguousarray(g).view(np.complex64).reshape(shape)


def sampler_noise(seed: int, n_draws: int, shape) -> np.ndarray:
    """[n_draws, *shape] complex64: draw k is consumed k-th by the sampler (prior first, then per step
    corrector draws followed by the predictor draw -- reference sampling/__init__.py:62-68)."""
    return np.stack([complex_normal(seed, f"draw{k}", shape) for k in range(n_draws)])


def synth_noisy_speech(n_utts: int, length: int, sr: int = 24000, seed: int = 1234) -> np.ndarray:
    """float32 [n_utts, length]: harmonic source with a random-walk f0 (90-250 Hz), 20 harmonics with
    1/k roll-off, 4 Hz syllabic envelope, white noise at 10 dB SNR, peak-normalised to 0.8
    (normalisation as the reference loader: data/components/loadwav_dataset.py:99-100;
    recipe: SURVEY.md section 8d)."""
    out = np.zeros((n_utts, length), dtype=np.float32)
    t = np.arange(length, dtype=np.float64) / sr
    for b in range(n_utts):
        nseg = length // 240 + 2                                   # f0 control points every 10 ms
        steps = (uniform01(seed + b, "f0walk", nseg) - 0.5) * 12.0
        f0c = 160.0 + np.cumsum(steps)
        f0c = 90.0 + np.abs(np.mod(f0c - 90.0, 320.0) - 160.0)      # reflect into [90, 250]
        f0 = np.interp(np.arange(length) / 240.0, np.arange(nseg), f0c)
        phase = 2.0 * np.pi * np.cumsum(f0) / sr
        sig = np.zeros(length, dtype=np.float64)
        for k in range(1, 21):
            sig += np.sin(k * phase + 0.37 * k) / k
        env = 0.55 + 0.45 * np.sin(2.0 * np.pi * 4.0 * t + 0.9 * b)
        sig *= env
        noise = normal(seed + b, "awgn", length).astype(np.float64)
        p_sig = np.mean(sig ** 2) + 1e-12
        noise *= np.sqrt(p_sig / (10.0 ** (10.0 / 10.0)) / (np.mean(noise ** 2) + 1e-12))
        x = sig + noise
        out[b] = (x / np.max(np.abs(x)) * 0.8).astype(np.float32)
    return out
