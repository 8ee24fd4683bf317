"""Spectrogram glue shared by ``ScoreModel`` and ``NCSNPP_Wrapper``: analysis / synthesis with a cached per-device window
(``torch.stft`` / ``torch.istft``, centred, periodic window) and the magnitude compression ``|S|^e e^{j arg S} * factor``.
Method names follow the reference classes (``model_wrapper.py:92-122`` in both the sgmse and the GAN generator package);
on CUDA tensors compression + frame padding and decompression each run as one kernel of libuse_hip.so.
"""
from __future__ import annotations

import torch

from .other import pad_spec

_WINDOWS = {"hann": lambda n: torch.hann_window(n, periodic=True),
            "sqrthann": lambda n: torch.hann_window(n, periodic=True).sqrt()}


def get_window(window_type, window_length):
    try:
        return _WINDOWS[window_type](window_length)
    except KeyError:
        raise NotImplementedError(f"Window type {window_type} not implemented!") from None


class SpectralGlue:
    """Mixin: call ``_init_spectral`` from ``__init__``."""

    def _init_spectral(self, n_fft, hop_length, num_frames, window, spec_factor, spec_abs_exponent):
        self.n_fft, self.hop_length, self.num_frames = n_fft, hop_length, num_frames
        self.spec_factor, self.spec_abs_exponent = spec_factor, spec_abs_exponent
        self.window = get_window(window, n_fft)
        self.windows = {}                                      # device -> window
        self.target_len = (num_frames - 1) * hop_length

    def _get_window(self, like):
        if like.device not in self.windows:
            self.windows[like.device] = self.window.to(like.device)
        return self.windows[like.device]

    def _fft_args(self, like):
        return dict(n_fft=self.n_fft, hop_length=self.hop_length, window=self._get_window(like), center=True)

    def stft(self, sig):
        return torch.stft(sig, return_complex=True, **self._fft_args(sig))

    def istft(self, spec, length=None):
        return torch.istft(spec, length=length, **self._fft_args(spec))

    @staticmethod
    def _power_law(spec, e):
        return spec if e == 1 else torch.polar(spec.abs() ** e, spec.angle())

    def spec_fwd(self, spec):
        return self._power_law(spec, self.spec_abs_exponent) * self.spec_factor

    def spec_back(self, spec):
        return self._power_law(spec / self.spec_factor, 1 / self.spec_abs_exponent)

    device_stft = True     # CUDA tensors: analysis / synthesis in libuse_hip.so (use_stft_fwd / use_istft_back); False: torch.stft / istft

    def _device_stft_ok(self, length):
        return self.device_stft and self.n_fft % 2 == 0 and length > self.n_fft // 2 and self.hop_length <= self.n_fft

    def _spectrogram(self, y):
        """waveform [B, L] -> compressed spectrogram [B, 1, F, T'] with T' padded to a multiple of 64 frames."""
        if y.is_cuda and y.dtype == torch.float32 and self._device_stft_ok(y.shape[1]):
            from ...hip_engine import stft_compress_pad
            return stft_compress_pad(y, self._get_window(y), self.n_fft, self.hop_length, self.spec_factor, self.spec_abs_exponent)
        S = self.stft(y)
        if S.is_cuda:
            from ...hip_engine import spec_compress_pad
            return spec_compress_pad(S, self.spec_factor, self.spec_abs_exponent)
        return pad_spec(self.spec_fwd(S).unsqueeze(1))

    def _waveform(self, X, length):
        """[B, 1, F, T'] -> waveform [B, length].  All T' frames enter the iSTFT, as in the reference: the frames of the
        padding region overlap the last n_fft/2 samples of the signal."""
        if X.is_cuda and length is not None and self._device_stft_ok(length) and X.shape[3] >= 1 + length // self.hop_length:
            from ...hip_engine import istft_decompress
            return istft_decompress(X, self._get_window(X), self.n_fft, self.hop_length, length, self.spec_factor, self.spec_abs_exponent)
        if X.is_cuda:
            from ...hip_engine import spec_decompress_crop
            return self.istft(spec_decompress_crop(X, X.shape[3], self.spec_factor, self.spec_abs_exponent), length)
        return self.istft(self.spec_back(X.squeeze(1)), length)
