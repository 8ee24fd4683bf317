"""``GANModule`` with the constructor keys and ``predict_step`` contract of the reference's
``src/models/LSGAN_module.py:10-49,139-155`` -- the refine stage of the reference's documented pipeline (SGMSE sampler, then
this): ``predict_step(batch, batch_idx)`` runs ``G(batch)``, trims every ``fake`` item to ``sample_length`` and writes it to
``audio_path.replace(data_folder, target_folder)``.  A ``lightning.LightningModule`` where Lightning is installed, a plain ``nn.Module``
otherwise (as ``SGMSEModule``).  Only the generator is served; discriminator, criteria and optimisers are accepted and ignored (GAN
training is out of scope).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from .SGMSE_module import _Base, _write_wav


class GANModule(_Base):
    def __init__(self, G: torch.nn.Module, D=None, G_optimizer=None, D_optimizer=None, G_scheduler=None, D_scheduler=None,
                 G_criterion=None, D_criterion=None, compile: bool = False, accumulate_grad_batches: int = 1,
                 rewrite_lr=False, G_lr=None, D_lr=None, wav_subtype: str = "PCM_16"):
        super().__init__()
        self.G = G
        self.wav_subtype = wav_subtype
        self.compile = compile

    def load_lightning_checkpoint(self, path: str, map_location="cpu"):
        """Loads the ``G.*`` tensors of ``ckpt['state_dict']`` (reference key layout ``G.net.all_modules...``)."""
        ckpt = torch.load(path, map_location=map_location, weights_only=False)
        sd = {k: v for k, v in ckpt.get("state_dict", ckpt).items() if k.startswith("G.")}
        missing, unexpected = self.load_state_dict(sd, strict=False)
        if missing:
            raise KeyError(f"checkpoint is missing {len(missing)} generator tensors, e.g. {missing[:3]}")
        return unexpected

    @torch.no_grad()
    def predict_step(self, batch: dict, batch_idx: int = 0) -> dict:
        batch = self.G(batch)
        for i, fake in enumerate(batch["fake"]):
            if "audio_path" not in batch:
                continue
            noisy_path = batch["audio_path"][i]
            sample_length = int(batch["sample_length"][i])
            sample_rate = batch["sampling_rate"][i]
            enhanced_path = noisy_path.replace(batch["data_folder"], batch["target_folder"])
            os.makedirs(os.path.dirname(enhanced_path) or ".", exist_ok=True)
            _write_wav(enhanced_path, fake.detach().cpu().numpy().astype(np.float32)[:sample_length], sample_rate, self.wav_subtype)
        return batch

    def training_step(self, *a, **k):
        raise NotImplementedError("training is outside the scope of the MI355X sampling library")
