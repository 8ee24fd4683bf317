"""``NCSNPP_Wrapper`` with the constructor and inference contract of the reference's
``src/models/components/GAN/generator/ncsnpp/model_wrapper.py:19-121``: STFT -> compress -> pad to a multiple of 64 frames ->
``NCSNpp(discriminative=True)`` (one network evaluation in libuse_hip.so) -> decompress -> iSTFT, adding ``batch["fake"]``.

The training branch (random crops of ``clean`` / ``perturbed`` pairs, reference lines 88-112) is outside the scope of the
library and raises; the state-dict keys (``net.all_modules...``, ``net.output_layer...``) are the reference's.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ..sgmse.backbones.ncsnpp import NCSNpp
from ..sgmse.util.other import pad_spec


def get_window(window_type, window_length):
    if window_type == "sqrthann":
        return torch.sqrt(torch.hann_window(window_length, periodic=True))
    if window_type == "hann":
        return torch.hann_window(window_length, periodic=True)
    raise NotImplementedError(f"Window type {window_type} not implemented!")


class NCSNPP_Wrapper(nn.Module):
    def __init__(self, n_fft=510, hop_length=128, num_frames=256, window="hann", spec_factor=0.15, spec_abs_exponent=0.5,
                 precision="bf16"):
        super().__init__()
        self.n_fft, self.hop_length, self.num_frames = n_fft, hop_length, num_frames
        self.window = get_window(window, n_fft)
        self.windows = {}
        self.spec_factor, self.spec_abs_exponent = spec_factor, spec_abs_exponent
        self.target_len = (num_frames - 1) * hop_length
        self.net = NCSNpp(discriminative=True, precision=precision)

    def spec_fwd(self, spec):
        if self.spec_abs_exponent != 1:
            spec = spec.abs() ** self.spec_abs_exponent * torch.exp(1j * spec.angle())
        return spec * self.spec_factor

    def spec_back(self, spec):
        spec = spec / self.spec_factor
        if self.spec_abs_exponent != 1:
            spec = spec.abs() ** (1 / self.spec_abs_exponent) * torch.exp(1j * spec.angle())
        return spec

    def _get_window(self, x):
        w = self.windows.get(x.device)
        if w is None:
            w = self.windows[x.device] = self.window.to(x.device)
        return w

    def stft(self, sig):
        return torch.stft(sig, n_fft=self.n_fft, hop_length=self.hop_length, window=self._get_window(sig), center=True,
                          return_complex=True)

    def istft(self, spec, length=None):
        return torch.istft(spec, n_fft=self.n_fft, hop_length=self.hop_length, window=self._get_window(spec), center=True,
                           length=length)

    @torch.no_grad()
    def forward(self, batch_data: dict) -> dict:
        if "clean" in batch_data:
            raise NotImplementedError("the training branch of NCSNPP_Wrapper is outside the scope of the MI355X library")
        y = batch_data["perturbed"]
        T_orig = y.size(1)
        S = self.stft(y)
        if S.is_cuda:                                         # fused compression + padding / decompression kernels
            from ..hip_engine import spec_compress_pad, spec_decompress_crop
            Y = self.net(spec_compress_pad(S, self.spec_factor, self.spec_abs_exponent))
            batch_data["fake"] = self.istft(spec_decompress_crop(Y, Y.shape[3], self.spec_factor, self.spec_abs_exponent), T_orig)
            return batch_data
        Y = pad_spec(self.spec_fwd(S).unsqueeze(1)).contiguous()
        Y = self.net(Y)
        batch_data["fake"] = self.istft(self.spec_back(Y.squeeze(1)), T_orig)
        return batch_data
