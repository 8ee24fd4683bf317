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
from ..sgmse.util.spectral import SpectralGlue, get_window  # noqa: F401


class NCSNPP_Wrapper(SpectralGlue, nn.Module):
    def __init__(self, n_fft=510, hop_length=128, num_frames=256, window="hann", spec_factor=0.15, spec_abs_exponent=0.5,
                 precision="bf16"):
        super().__init__()
        self._init_spectral(n_fft, hop_length, num_frames, window, spec_factor, spec_abs_exponent)
        self.net = NCSNpp(discriminative=True, precision=precision)

    @torch.no_grad()
    def forward(self, batch_data: dict) -> dict:
        if "clean" in batch_data:
            raise NotImplementedError("the training branch of NCSNPP_Wrapper is outside the scope of the MI355X library")
        noisy = batch_data["perturbed"]
        refined = self.net(self._spectrogram(noisy).contiguous())
        batch_data["fake"] = self._waveform(refined, noisy.size(1))
        return batch_data
