"""ORACLE (test infrastructure, not product code) -- CPU fp32 restatement of the NCSN++ score network.

This file restates, in plain functional PyTorch on CPU tensors, the forward pass that the HIP
library implements.  It is imported only by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; nothing under ``universal_speech_enhancement_amd/`` may import it.

Pinning: the reference has no tests or golden vectors for this path, so the
restatement is pinned against outputs of the reference itself, produced in the build container by
``oracle/gen_golden.py`` (which imports /root/reference) and committed under ``tests/golden/``;
``tests/test_oracle_golden.py`` re-checks it on every run without the reference.

Each function cites the reference lines it follows (paths relative to
``/root/reference/src/models/components/sgmse/``).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


def _fir_kernel(k: Sequence[float] = (1, 3, 3, 1), gain: float = 1.0) -> torch.Tensor:
    """backbones/ncsnpp_utils/up_or_down_sampling.py:188-195 (_setup_kernel): outer product, normalised."""
    k1 = np.asarray(k, dtype=np.float32)
    k2 = np.outer(k1, k1)
    k2 /= np.sum(k2)
    return torch.from_numpy((k2 * gain).astype(np.float32))


def fir_upsample2(x: torch.Tensor, k: Sequence[float] = (1, 3, 3, 1)) -> torch.Tensor:
    """upsample_2d(x, k, factor=2) (up_or_down_sampling.py:202-232) through upfirdn2d_native
    (op/upfirdn2d.py:173-208): zero-stuff by 2, pad (2, 1), correlate with the flipped 4x4 kernel
    k*factor**2.  Written here as the explicit zero-stuff + pad + depthwise conv."""
    B, C, H, W = x.shape
    kern = _fir_kernel(k, gain=4.0)
    z = x.new_zeros(B, C, H, 2, W, 2)
    z[:, :, :, 0, :, 0] = x
    z = z.reshape(B, C, 2 * H, 2 * W)
    z = F.pad(z, (2, 1, 2, 1))
    w = torch.flip(kern, [0, 1]).view(1, 1, 4, 4).expand(C, 1, 4, 4).to(x.dtype)
    return F.conv2d(z, w, groups=C)


def fir_downsample2(x: torch.Tensor, k: Sequence[float] = (1, 3, 3, 1)) -> torch.Tensor:
    """downsample_2d(x, k, factor=2) (up_or_down_sampling.py:235-264): pad (1, 1), correlate with
    the flipped 4x4 kernel, keep every 2nd sample (op/upfirdn2d.py:186-203)."""
    B, C, H, W = x.shape
    kern = _fir_kernel(k, gain=1.0)
    z = F.pad(x, (1, 1, 1, 1))
    w = torch.flip(kern, [0, 1]).view(1, 1, 4, 4).expand(C, 1, 4, 4).to(x.dtype)
    return F.conv2d(z, w, groups=C, stride=2)


def _gn(x: torch.Tensor, sd: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """nn.GroupNorm(min(C//4, 32), C, eps=1e-6) (layerspp.py:255-257, ncsnpp.py:269-271)."""
    C = x.shape[1]
    return F.group_norm(x, min(C // 4, 32), sd[prefix + ".weight"], sd[prefix + ".bias"], eps=1e-6)


def _nin(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """NIN (layers.py:639-650): per-pixel x[b,h,w,:] @ W[C,C'] + b."""
    y = torch.einsum("bchw,cd->bdhw", x, W)
    return y + b[None, :, None, None]


def resblock_biggan(x, temb, sd, p, up=False, down=False):
    """ResnetBlockBigGANpp.forward (layerspp.py:282-314); act = SiLU (layers.py:38-39);
    Dropout(p=0.0) in eval is the identity; skip_rescale=True (ncsnpp.py:54)."""
    h = F.silu(_gn(x, sd, p + ".GroupNorm_0"))
    if up:
        h = fir_upsample2(h); x = fir_upsample2(x)
    elif down:
        h = fir_downsample2(h); x = fir_downsample2(x)
    h = F.conv2d(h, sd[p + ".Conv_0.weight"], sd[p + ".Conv_0.bias"], padding=1)
    if temb is not None:
        h = h + F.linear(F.silu(temb), sd[p + ".Dense_0.weight"], sd[p + ".Dense_0.bias"])[:, :, None, None]
    h = F.silu(_gn(h, sd, p + ".GroupNorm_1"))
    h = F.conv2d(h, sd[p + ".Conv_1.weight"], sd[p + ".Conv_1.bias"], padding=1)
    if (p + ".Conv_2.weight") in sd:
        x = F.conv2d(x, sd[p + ".Conv_2.weight"], sd[p + ".Conv_2.bias"])
    return (x + h) / SQRT2


def attn_block(x, sd, p):
    """AttnBlockpp.forward (layerspp.py:77-93), skip_rescale=True."""
    B, C, H, W = x.shape
    h = _gn(x, sd, p + ".GroupNorm_0")
    q = _nin(h, sd[p + ".NIN_0.W"], sd[p + ".NIN_0.b"])
    k = _nin(h, sd[p + ".NIN_1.W"], sd[p + ".NIN_1.b"])
    v = _nin(h, sd[p + ".NIN_2.W"], sd[p + ".NIN_2.b"])
    w = torch.einsum("bchw,bcij->bhwij", q, k) * (int(C) ** (-0.5))
    w = F.softmax(w.reshape(B, H, W, H * W), dim=-1).reshape(B, H, W, H, W)
    h = torch.einsum("bhwij,bcij->bchw", w, v)
    h = _nin(h, sd[p + ".NIN_3.W"], sd[p + ".NIN_3.b"])
    return (x + h) / SQRT2


def time_embedding(t: torch.Tensor, sd) -> torch.Tensor:
    """GaussianFourierProjection(log t) (layerspp.py:37-39, ncsnpp.py:351-352) then
    Linear -> SiLU -> Linear (ncsnpp.py:364-368)."""
    xp = torch.log(t)[:, None] * sd["all_modules.0.W"][None, :] * 2 * np.pi
    e = torch.cat([torch.sin(xp), torch.cos(xp)], dim=-1)
    e = F.linear(e, sd["all_modules.1.weight"], sd["all_modules.1.bias"])
    return F.linear(F.silu(e), sd["all_modules.2.weight"], sd["all_modules.2.bias"])


def ncsnpp_forward(
    sd: Dict[str, torch.Tensor],
    x: torch.Tensor,
    t: torch.Tensor,
    ch_mult: Sequence[int] = (1, 1, 2, 2, 2, 2, 2),
    num_res_blocks: int = 2,
    taps: Optional[dict] = None,
    discriminative: bool = False,
) -> torch.Tensor:
    """NCSNpp.forward (ncsnpp.py:324-501) for the predict-path configuration
    (biggan blocks, fir, output_skip / input_skip 'sum', fourier, scale_by_sigma, not centered).

    x: complex64 [B, 2, F, T] = cat([x_t, Y], dim=1); t: float32 [B].  Returns complex64 [B, 1, F, T].
    ``taps`` (optional dict) receives named intermediates for per-layer parity tests.

    ``discriminative=True`` (ncsnpp.py:86-92; the generator of the LSGAN refine stage,
    GAN/generator/ncsnpp/model_wrapper.py:54,114-121): x is complex64 [B, 1, F, T] alone, ``t`` is ignored -- no time
    embedding (the two Linear layers are absent from the module list, Dense_0 is unused), no division by t.
    """
    L = len(ch_mult)
    if discriminative:
        x4 = torch.cat([x[:, [0]].real, x[:, [0]].imag], dim=1)               # ncsnpp.py:333-347 with input_channels = 2
        temb = None                                                           # ncsnpp.py:352, 364-370
        m = 1
    else:
        # ncsnpp.py:333-347: channels = (x.re, x.im, y.re, y.im) [+ (y2.re, y2.im): input_channels = 6, condition="both"]
        x4 = torch.cat([p for k in range(x.shape[1]) for p in (x[:, [k]].real, x[:, [k]].imag)], dim=1)
        temb = time_embedding(t, sd)
        m = 3
    x4 = 2 * x4 - 1.0  # ncsnpp.py:372-374
    input_pyramid = x4
    hs: List[torch.Tensor] = [F.conv2d(x4, sd[f"all_modules.{m}.weight"], sd[f"all_modules.{m}.bias"], padding=1)]
    m += 1
    if taps is not None:
        taps["temb"] = temb; taps["h_in"] = hs[0]
    for lvl in range(L):
        for _ in range(num_res_blocks):
            h = resblock_biggan(hs[-1], temb, sd, f"all_modules.{m}"); m += 1
            hs.append(h)
        if lvl != L - 1:
            h = resblock_biggan(hs[-1], temb, sd, f"all_modules.{m}", down=True); m += 1
            input_pyramid = fir_downsample2(input_pyramid)                     # ncsnpp.py:404
            h = F.conv2d(input_pyramid, sd[f"all_modules.{m}.Conv_0.weight"],  # Combine 'sum', layerspp.py:50-55
                         sd[f"all_modules.{m}.Conv_0.bias"]) + h
            m += 1
            hs.append(h)
    h = hs[-1]
    h = resblock_biggan(h, temb, sd, f"all_modules.{m}"); m += 1
    if taps is not None:
        taps["pre_attn"] = h
    h = attn_block(h, sd, f"all_modules.{m}"); m += 1
    if taps is not None:
        taps["post_attn"] = h
    h = resblock_biggan(h, temb, sd, f"all_modules.{m}"); m += 1
    pyramid = None
    for lvl in reversed(range(L)):
        for _ in range(num_res_blocks + 1):
            h = resblock_biggan(torch.cat([h, hs.pop()], dim=1), temb, sd, f"all_modules.{m}"); m += 1
        ph = F.silu(_gn(h, sd, f"all_modules.{m}")); m += 1                    # ncsnpp.py:443,457
        ph = F.conv2d(ph, sd[f"all_modules.{m}.weight"], sd[f"all_modules.{m}.bias"], padding=1); m += 1
        pyramid = ph if pyramid is None else fir_upsample2(pyramid) + ph      # ncsnpp.py:456-461
        if lvl != 0:
            h = resblock_biggan(h, temb, sd, f"all_modules.{m}", up=True); m += 1
    assert not hs
    if taps is not None:
        taps["pyramid"] = pyramid
    h = pyramid if discriminative else pyramid / t[:, None, None, None]        # ncsnpp.py:492-494
    h = F.conv2d(h, sd["output_layer.weight"], sd["output_layer.bias"])        # ncsnpp.py:497
    return torch.complex(h[:, 0], h[:, 1]).unsqueeze(1)                        # ncsnpp.py:498-500


def to_torch(sd_np: Dict[str, np.ndarray]) -> Dict[str, torch.Tensor]:
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd_np.items()}
