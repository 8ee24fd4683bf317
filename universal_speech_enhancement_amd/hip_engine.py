"""Thin Python owner of one ``use_handle`` (include/use_hip.h): uploads weights, plans the workspace for a
(B, T') shape and forwards torch CUDA tensors' raw pointers and the current HIP stream.

PyTorch is plumbing only here (device memory, streams); all arithmetic on the path runs in libuse_hip.so.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import UseConfig, UseHipError, UseSamplerConfig, check


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _require_cuda_c64(name: str, t: torch.Tensor, shape=None) -> torch.Tensor:
    if not t.is_cuda:
        raise UseHipError(f"{name} must be a CUDA (ROCm) tensor: the sampling path has no CPU implementation")
    if t.dtype != torch.complex64:
        raise TypeError(f"{name} must be complex64, got {t.dtype}")
    t = t.contiguous()
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


def set_option(name: str, value: int) -> None:
    """Process-wide tuning knob of the library (``use_set_option``), e.g. ``conv_v4_min_blocks``."""
    check(_lib.lib().use_set_option(name.encode(), int(value)), "use_set_option")


def spec_compress_pad(stft: torch.Tensor, factor: float, exponent: float, multiple: int = 64) -> torch.Tensor:
    """``pad_spec(spec_fwd(stft).unsqueeze(1))`` in one kernel (``use_spec_fwd``): complex64 CUDA [B,F,T] -> [B,1,F,T']."""
    if not stft.is_cuda or stft.dtype != torch.complex64 or stft.dim() != 3:
        raise UseHipError("spec_compress_pad needs a complex64 CUDA tensor [B, F, T]")
    stft = stft.contiguous()
    B, F, T = stft.shape
    Tp = (T + multiple - 1) // multiple * multiple
    Y = torch.empty((B, 1, F, Tp), dtype=torch.complex64, device=stft.device)
    check(_lib.lib().use_spec_fwd(stft.data_ptr(), Y.data_ptr(), B, F, T, Tp, float(factor), float(exponent), _stream_ptr(stft.device)),
          "use_spec_fwd")
    return Y


def spec_decompress_crop(X: torch.Tensor, T: int, factor: float, exponent: float) -> torch.Tensor:
    """``spec_back(X.squeeze(1))[..., :T]`` in one kernel (``use_spec_back``): complex64 CUDA [B,1,F,T'] -> [B,F,T]."""
    if not X.is_cuda or X.dtype != torch.complex64 or X.dim() != 4 or X.shape[1] != 1:
        raise UseHipError("spec_decompress_crop needs a complex64 CUDA tensor [B, 1, F, T']")
    X = X.contiguous()
    B, _, F, Tp = X.shape
    S = torch.empty((B, F, T), dtype=torch.complex64, device=X.device)
    check(_lib.lib().use_spec_back(X.data_ptr(), S.data_ptr(), B, F, int(T), Tp, float(factor), float(exponent), _stream_ptr(X.device)),
          "use_spec_back")
    return S


def stft_compress_pad(wav: torch.Tensor, window: torch.Tensor, n_fft: int, hop: int, factor: float, exponent: float,
                      multiple: int = 64) -> torch.Tensor:
    """``pad_spec(spec_fwd(stft(wav)).unsqueeze(1))`` in one kernel (``use_stft_fwd``): float32 CUDA [B, L] -> complex64 [B,1,F,T']."""
    if not wav.is_cuda or wav.dtype != torch.float32 or wav.dim() != 2:
        raise UseHipError("stft_compress_pad needs a float32 CUDA tensor [B, L]")
    wav, window = wav.contiguous(), window.to(device=wav.device, dtype=torch.float32).contiguous()
    B, L = wav.shape
    T = 1 + L // hop
    Tp = (T + multiple - 1) // multiple * multiple
    Y = torch.empty((B, 1, n_fft // 2 + 1, Tp), dtype=torch.complex64, device=wav.device)
    check(_lib.lib().use_stft_fwd(wav.data_ptr(), Y.data_ptr(), B, L, int(n_fft), int(hop), window.data_ptr(), Tp, float(factor),
                                  float(exponent), _stream_ptr(wav.device)), "use_stft_fwd")
    return Y


def istft_decompress(X: torch.Tensor, window: torch.Tensor, n_fft: int, hop: int, length: int, factor: float,
                     exponent: float) -> torch.Tensor:
    """``istft(spec_back(X.squeeze(1)), length)`` in one kernel (``use_istft_back``): complex64 CUDA [B,1,F,T'] -> float32 [B, length]."""
    if not X.is_cuda or X.dtype != torch.complex64 or X.dim() != 4 or X.shape[1] != 1 or X.shape[2] != n_fft // 2 + 1:
        raise UseHipError("istft_decompress needs a complex64 CUDA tensor [B, 1, n_fft/2+1, T']")
    X, window = X.contiguous(), window.to(device=X.device, dtype=torch.float32).contiguous()
    B, _, _, Tp = X.shape
    wav = torch.empty((B, int(length)), dtype=torch.float32, device=X.device)
    check(_lib.lib().use_istft_back(X.data_ptr(), wav.data_ptr(), B, int(length), int(n_fft), int(hop), window.data_ptr(), Tp,
                                    float(factor), float(exponent), _stream_ptr(X.device)), "use_istft_back")
    return wav


class HipScoreEngine:
    """One handle per (process, device).  Not re-entrant."""

    def __init__(self, nf=128, ch_mult: Sequence[int] = (1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, n_freq=512,
                 precision="bf16", device: Optional[int] = None, theta=1.5, sigma_min=0.05, sigma_max=0.5,
                 input_channels=4, conditional=True, scale_by_sigma=True):
        if precision not in _lib.PREC:
            raise ValueError(f"precision must be one of {list(_lib.PREC)}, got {precision!r}")
        self.L = _lib.lib()
        self.device = torch.cuda.current_device() if device is None else int(device)
        cfg = UseConfig()
        cfg.nf, cfg.n_levels, cfg.num_res_blocks, cfg.n_freq = nf, len(ch_mult), num_res_blocks, n_freq
        for i, m in enumerate(ch_mult):
            cfg.ch_mult[i] = int(m)
        cfg.precision = _lib.PREC[precision]
        cfg.theta, cfg.sigma_min, cfg.sigma_max = theta, sigma_min, sigma_max
        cfg.input_channels, cfg.unconditional, cfg.no_sigma_scale = int(input_channels), int(not conditional), int(not scale_by_sigma)
        self.input_channels, self.conditional = int(input_channels), bool(conditional)
        self.cfg, self.precision, self.n_freq = cfg, precision, n_freq
        h = C.c_void_p()
        check(self.L.use_create(C.byref(cfg), self.device, C.byref(h)), "use_create")
        self.h = h
        self.plan_shape = None
        self.sampler_key = None
        self.weights_ready = False

    def close(self):
        if getattr(self, "h", None):
            self.L.use_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights ------------------------------------------------------------------------------------
    def save_weight_blob(self, path: str) -> None:
        """Packed weight file (``use_save_weight_blob``): after ``set_weights`` of every tensor (no GPU needed) or a commit."""
        check(self.L.use_save_weight_blob(self.h, os.fsencode(path)), "use_save_weight_blob")

    def load_weight_blob(self, path: str) -> None:
        """Start from a packed weight file instead of a state dict (``use_load_weight_blob``)."""
        check(self.L.use_load_weight_blob(self.h, os.fsencode(path)), "use_load_weight_blob")
        self.weights_ready = True
        self.sampler_key = None                               # the library dropped its time-embedding table and graphs

    def expected_weights(self) -> Dict[str, tuple]:
        out = {}
        n = check(self.L.use_num_expected_weights(self.h))
        for i in range(n):
            name, shape, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
            check(self.L.use_expected_weight(self.h, i, C.byref(name), shape, C.byref(nd)))
            out[name.value.decode()] = tuple(shape[: nd.value])
        return out

    def load_state_dict(self, sd: Dict[str, "np.ndarray | torch.Tensor"], prefix: str = ""):
        """Upload weights given under the reference's state-dict keys (optionally behind ``prefix``)."""
        self.set_weights(sd, prefix)
        check(self.L.use_commit_weights(self.h), "use_commit_weights")
        self.weights_ready = True
        self.sampler_key = None

    def set_weights(self, sd: Dict[str, "np.ndarray | torch.Tensor"], prefix: str = ""):
        """``use_set_weight`` for every tensor, without committing to the device (enough for ``save_weight_blob``)."""
        for name in self.expected_weights():
            key = prefix + name
            if key not in sd:
                raise KeyError(f"state dict is missing '{key}'")
            v = sd[key]
            a = v.detach().cpu().float().contiguous().numpy() if isinstance(v, torch.Tensor) else np.ascontiguousarray(v, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            check(self.L.use_set_weight(self.h, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim), f"use_set_weight({name})")

    def weight_blob(self) -> torch.Tensor:
        """uint8 CUDA view of the packed device blob (for a one-time RCCL broadcast from rank 0)."""
        p, n = C.c_void_p(), C.c_size_t()
        check(self.L.use_weight_blob(self.h, C.byref(p), C.byref(n)))
        if not p.value:
            raise UseHipError("weight blob not allocated (call load_state_dict or alloc_weight_blob first)")

        class _Iface:  # __cuda_array_interface__ shim: lets torch wrap library-owned device memory
            pass
        o = _Iface()
        o.__cuda_array_interface__ = {"shape": (n.value,), "typestr": "|u1", "data": (p.value, False), "version": 2}
        return torch.as_tensor(o, device=f"cuda:{self.device}")

    def alloc_weight_blob(self):
        check(self.L.use_alloc_weight_blob(self.h), "use_alloc_weight_blob")
        self.weights_ready = True
        self.sampler_key = None

    # ---- planning -----------------------------------------------------------------------------------
    def plan(self, B: int, Tpad: int):
        # (options are read at use_plan: after a use_set_option the C entry points refuse the stale plan with USE_E_STATE - re-plan here)
        if self.plan_shape != (B, Tpad) or self.stat("plan_stale"):
            check(self.L.use_plan(self.h, B, Tpad), "use_plan")
            self.plan_shape = (B, Tpad)
            self.sampler_key = None

    def workspace_bytes(self) -> int:
        n = C.c_size_t()
        check(self.L.use_workspace_bytes(self.h, C.byref(n)))
        return n.value

    def flops_per_score(self) -> float:
        return float(self.L.use_flops_per_score(self.h))

    # ---- execution ----------------------------------------------------------------------------------
    def score(self, x: torch.Tensor, y: torch.Tensor, t: torch.Tensor, y2: Optional[torch.Tensor] = None) -> torch.Tensor:
        """-score_net(cat[x, y], t): x, y complex64 [B,1,F,T'] on this device, t float32 [B].  ``y2``: the second conditioning
        spectrogram of a 6-channel network (condition="both"; ``use_score2``)."""
        x = _require_cuda_c64("x", x)
        y = _require_cuda_c64("y", y, x.shape)
        if y2 is not None:
            y2 = _require_cuda_c64("y2", y2, x.shape)
        B, _, Fq, T = x.shape
        if Fq != self.n_freq:
            raise ValueError(f"expected {self.n_freq} frequency bins, got {Fq}")
        self.plan(B, T)
        t = t.to(device=x.device, dtype=torch.float32).contiguous()
        if t.numel() != B:
            raise ValueError(f"t must have {B} elements")
        out = torch.empty_like(x)
        if y2 is not None:
            check(self.L.use_score2(self.h, x.data_ptr(), y.data_ptr(), y2.data_ptr(), t.data_ptr(), out.data_ptr(), _stream_ptr(x.device)),
                  "use_score2")
        else:
            check(self.L.use_score(self.h, x.data_ptr(), y.data_ptr(), t.data_ptr(), out.data_ptr(), _stream_ptr(x.device)), "use_score")
        return out

    def forward(self, x: torch.Tensor, y: Optional[torch.Tensor] = None, t: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Raw backbone output ``NCSNpp.forward(cat[x, y], t)`` (``use_forward``).  ``y`` is None for a 2-channel
        (discriminative) network, ``t`` is None for an unconditional one."""
        x = _require_cuda_c64("x", x)
        if y is not None:
            y = _require_cuda_c64("y", y, x.shape)
        self.plan(x.shape[0], x.shape[3])
        if t is not None:
            t = t.to(device=x.device, dtype=torch.float32).contiguous()
            if t.shape != (x.shape[0],):
                raise ValueError(f"t must have shape [{x.shape[0]}]")
        out = torch.empty_like(x)
        check(self.L.use_forward(self.h, x.data_ptr(), None if y is None else y.data_ptr(), None if t is None else t.data_ptr(),
                                 out.data_ptr(), _stream_ptr(x.device)), "use_forward")
        return out

    def profile_score(self, x, y, t):
        """(conv_ms, conv_flops, conv_bytes, conv_launches, total_ms) of one eager score evaluation: HIP events around every
        launch of the dominant kernel (conv_v4_kernel), its algorithmic FLOPs / HBM bytes, and the evaluation's wall time."""
        x = _require_cuda_c64("x", x); y = _require_cuda_c64("y", y, x.shape)
        self.plan(x.shape[0], x.shape[3])
        t = t.to(device=x.device, dtype=torch.float32).contiguous()
        out = torch.empty_like(x)
        ms, fl, by, n, tot = C.c_double(), C.c_double(), C.c_double(), C.c_int(), C.c_double()
        check(self.L.use_profile_score(self.h, x.data_ptr(), y.data_ptr(), t.data_ptr(), out.data_ptr(), _stream_ptr(x.device),
                                       C.byref(ms), C.byref(fl), C.byref(by), C.byref(n), C.byref(tot)), "use_profile_score")
        return ms.value, fl.value, by.value, n.value, tot.value

    def stat(self, name: str) -> int:
        """A counter of the handle (``use_get_stat``): "graph_captures", "plans_built", "plan_cache_hits", "plans_parked"."""
        v = C.c_longlong()
        check(self.L.use_get_stat(self.h, name.encode(), C.byref(v)), "use_get_stat")
        return int(v.value)

    def profile_aux(self, with_flops: bool = False):
        """The other kernels of the last ``profile_score``: [(kernel class, H, W, algorithmic bytes, ms[, algorithmic FLOPs])] in launch
        order - the HBM-bound classes (fir_up, fir_down, pyr_conv, conv_in: FLOPs 0) and the MFMA-bound ones beside the dominant kernel
        (conv_v2, conv_sk)."""
        out, i = [], 0
        name = C.create_string_buffer(32)
        H, W, by, ms = C.c_int(), C.c_int(), C.c_double(), C.c_double()
        while True:
            rc = self.L.use_profile_aux(self.h, i, name, 32, C.byref(H), C.byref(W), C.byref(by), C.byref(ms))
            if rc == 1:
                return out
            check(rc, "use_profile_aux")
            fl = C.c_double()
            check(self.L.use_profile_aux_flops(self.h, i, C.byref(fl)), "use_profile_aux_flops")
            out.append((name.value.decode(), H.value, W.value, by.value, ms.value) + ((fl.value,) if with_flops else ()))
            i += 1

    def set_sampler(self, N, predictor="reverse_diffusion", corrector="none", corrector_steps=1, snr=0.5, t_eps=3e-2,
                    use_graph=True):
        key = (self.plan_shape, N, predictor, corrector, corrector_steps, float(snr), float(t_eps), bool(use_graph))
        if key == self.sampler_key:
            return
        sc = UseSamplerConfig(int(N), _lib.PREDICTORS[predictor], _lib.CORRECTORS[corrector], int(corrector_steps),
                              float(snr), float(t_eps), int(bool(use_graph)))
        check(self.L.use_set_sampler(self.h, C.byref(sc)), "use_set_sampler")
        self.sampler_key = key

    def num_noise_draws(self) -> int:
        return check(self.L.use_num_noise_draws(self.h))

    def timesteps(self) -> np.ndarray:
        n = check(self.L.use_get_timesteps(self.h, (C.c_float * 1)(), 0))
        buf = (C.c_float * n)()
        check(self.L.use_get_timesteps(self.h, buf, n))
        return np.array(buf[:], dtype=np.float32)

    def sample(self, y: torch.Tensor, noise: Optional[torch.Tensor] = None, seed: int = 0,
               cond: Optional[torch.Tensor] = None, cond2: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Run the configured PC sampler on y (complex64 [B,1,F,T']); returns x_mean of the last step.  ``cond``: score
        conditioning when it is not y itself (``use_sample_cond``); ``cond2``: the second conditioning spectrogram of a
        6-channel network (``use_sample_cond2``)."""
        y = _require_cuda_c64("y", y)
        cptr = None if cond is None or cond is y else _require_cuda_c64("cond", cond, y.shape).data_ptr()
        c2ptr = None if cond2 is None else _require_cuda_c64("cond2", cond2, y.shape).data_ptr()
        if (y.shape[0], y.shape[3]) != self.plan_shape:
            raise UseHipError(f"sampler planned for {self.plan_shape}, got B={y.shape[0]} T'={y.shape[3]}")
        nptr = None
        if noise is not None:
            noise = _require_cuda_c64("noise", noise, (self.num_noise_draws(),) + tuple(y.shape))
            nptr = noise.data_ptr()
        out = torch.empty_like(y)
        check(self.L.use_sample_cond2(self.h, y.data_ptr(), cptr, c2ptr, nptr, int(seed) & (2**64 - 1), out.data_ptr(),
                                      _stream_ptr(y.device)), "use_sample_cond2")
        return out

    def fill_noise(self, seed: int, draw: int, shape) -> torch.Tensor:
        """Draw ``draw`` of the device noise stream of ``sample(noise=None, seed=seed)`` as a complex64 tensor of ``shape`` = the whole
        batch tensor [B,1,F,T'] (``use_fill_noise``): draw 0 is the prior's, then per reverse step the corrector draws and the
        predictor draw.  Replaying these through ``sample(noise=...)`` reproduces the device-noise run bit for bit."""
        out = torch.empty(tuple(shape), dtype=torch.complex64, device=f"cuda:{self.device}")
        check(self.L.use_fill_noise(self.h, int(seed) & (2**64 - 1), int(draw), out.data_ptr(), out.numel(), _stream_ptr(out.device)), "use_fill_noise")
        return out

    def debug_tensor(self, name: str) -> torch.Tensor:
        """Copy of a named intermediate of the last score evaluation as float32 [B,H,W,C]."""
        p, dims, dt = C.c_void_p(), (C.c_int * 4)(), C.c_int()
        check(self.L.use_debug_tensor(self.h, name.encode(), C.byref(p), dims, C.byref(dt)))
        n = int(np.prod(dims[:]))

        class _Iface:
            pass
        o = _Iface()
        o.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f4" if dt.value == 0 else "<u2", "data": (p.value, False), "version": 2}
        raw = torch.as_tensor(o, device=f"cuda:{self.device}").clone()
        if dt.value == 1:
            raw = raw.view(torch.bfloat16)
        elif dt.value == 2:
            raw = raw.view(torch.float16)
        return raw.float().view(*dims[:])

    # ---- stand-alone SDE element-wise updates (use_sde_*) ----------------------------------------------
    def sde_prior(self, y, noise=None, seed=0):
        y = _require_cuda_c64("y", y)
        x = torch.empty_like(y)
        check(self.L.use_sde_prior(self.h, y.data_ptr(), None if noise is None else _require_cuda_c64("noise", noise, y.shape).data_ptr(),
                                   int(seed), x.data_ptr(), y.numel(), _stream_ptr(y.device)), "use_sde_prior")
        return x

    def sde_predictor(self, predictor, t: float, N: int, x, y, score, noise=None, seed=0):
        x = _require_cuda_c64("x", x)
        y = _require_cuda_c64("y", y, x.shape); score = _require_cuda_c64("score", score, x.shape)
        xo, xm = torch.empty_like(x), torch.empty_like(x)
        check(self.L.use_sde_predictor(self.h, _lib.PREDICTORS[predictor], float(t), int(N), x.data_ptr(), y.data_ptr(), score.data_ptr(),
                                       None if noise is None else _require_cuda_c64("noise", noise, x.shape).data_ptr(), int(seed),
                                       xo.data_ptr(), xm.data_ptr(), x.numel(), _stream_ptr(x.device)), "use_sde_predictor")
        return xo, xm

    def sde_corrector(self, corrector, t: float, snr: float, x, score, noise=None, seed=0):
        x = _require_cuda_c64("x", x)
        score = _require_cuda_c64("score", score, x.shape)
        xo, xm = torch.empty_like(x), torch.empty_like(x)
        check(self.L.use_sde_corrector(self.h, _lib.CORRECTORS[corrector], float(t), float(snr), x.shape[0], x.data_ptr(), score.data_ptr(),
                                       None if noise is None else _require_cuda_c64("noise", noise, x.shape).data_ptr(), int(seed),
                                       xo.data_ptr(), xm.data_ptr(), x.numel(), _stream_ptr(x.device)), "use_sde_corrector")
        return xo, xm
