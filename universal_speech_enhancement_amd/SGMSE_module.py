"""``SGMSEModule`` with the constructor and step contract of the reference's ``src/models/SGMSE_module.py:10-82``:
``predict_step(batch, batch_idx)`` runs ``Score.sample(batch)``, trims every item to ``sample_length`` and writes it to
``audio_path.replace(data_folder, target_folder)``; ``training_step`` / ``validation_step`` / ``test_step`` /
``configure_optimizers`` as there.  The base class is ``lightning.LightningModule`` where Lightning is installed (the reference's
``Trainer.fit`` / ``Trainer.predict`` then drive it unchanged, with the ``self.log`` calls of the reference) and a plain ``nn.Module``
where it is not (the target image) - the methods are the same either way.
"""
from __future__ import annotations

import os

import numpy as np
import torch


def _write_wav(path: str, wav: np.ndarray, sr: int, subtype: str = "PCM_16"):
    """``sf.write(path, wav, sr)`` of the reference (SGMSE_module.py:80): soundfile's default WAV subtype is 16-bit PCM;
    ``subtype="FLOAT"`` keeps the float32 samples.  Native writer (csrc/use_io.cpp: use_wav_write)."""
    from .wavio import FLOAT32, PCM16, write_wav
    if subtype not in ("PCM_16", "FLOAT"):
        raise ValueError(f"unsupported WAV subtype {subtype!r} (PCM_16 or FLOAT)")
    write_wav(path, wav, int(sr), PCM16 if subtype == "PCM_16" else FLOAT32)


try:                                                     # reference: class SGMSEModule(LightningModule), SGMSE_module.py:10
    from lightning import LightningModule as _Base
    HAS_LIGHTNING = True
except Exception:                                        # not installed on the target image
    _Base, HAS_LIGHTNING = torch.nn.Module, False


class SGMSEModule(_Base):
    def __init__(self, Score: torch.nn.Module, optimizer=None, scheduler=None, compile: bool = False, sampler_kwargs=None,
                 wav_subtype: str = "PCM_16"):
        super().__init__()
        self.Score = Score
        self.wav_subtype = wav_subtype                       # "PCM_16" = what the reference's sf.write produces; "FLOAT" = float32
        self.optimizer, self.scheduler, self.compile = optimizer, scheduler, compile
        self.sampler_kwargs = dict(sampler_kwargs or {})     # optional N / corrector_steps / snr overrides

    def load_lightning_checkpoint(self, path: str, map_location="cpu"):
        """Loads ``ckpt['state_dict']`` with the reference's key layout (``Score.score_net.all_modules...``)."""
        ckpt = torch.load(path, map_location=map_location, weights_only=False)
        sd = ckpt.get("state_dict", ckpt)
        missing, unexpected = self.load_state_dict(sd, strict=False)
        if missing:
            raise KeyError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[:3]}")
        return unexpected

    @torch.no_grad()
    def predict_step(self, batch: dict, batch_idx: int = 0) -> dict:
        batch = self.Score.sample(batch, **self.sampler_kwargs)
        for i, enhanced in enumerate(batch["enhanced"]):
            if "audio_path" not in batch:
                continue
            noisy_path = batch["audio_path"][i]
            sample_length = int(batch["sample_length"][i])
            sample_rate = batch["sampling_rate"][i]
            enhanced_path = noisy_path.replace(batch["data_folder"], batch["target_folder"])
            os.makedirs(os.path.dirname(enhanced_path) or ".", exist_ok=True)
            wav = enhanced.detach().cpu().numpy().astype(np.float32)[:sample_length]
            _write_wav(enhanced_path, wav, sample_rate, self.wav_subtype)
        return batch

    def get_score_loss(self, batch: dict) -> torch.Tensor:
        """Reference :42-44."""
        return self.Score.train_step(batch)

    def _log(self, name, value, **kw):
        """``self.log`` of the reference's steps when a Lightning trainer is attached; nothing otherwise."""
        if HAS_LIGHTNING and getattr(self, "_trainer", None) is not None:
            self.log(name, value, **kw)

    @torch.no_grad()
    def validation_step(self, batch: dict, batch_idx: int = 0) -> torch.Tensor:
        """Reference :56-58 (logs ``val/loss_Score``; the value is also returned)."""
        loss = self.get_score_loss(batch)
        self._log("val/loss_Score", loss, on_step=True, on_epoch=True, prog_bar=True)
        return loss

    @torch.no_grad()
    def test_step(self, batch: dict, batch_idx: int = 0) -> torch.Tensor:
        """Reference :61-63."""
        loss = self.get_score_loss(batch)
        self._log("test/loss_Score", loss, on_step=True, on_epoch=True, prog_bar=True)
        return loss

    def training_step(self, batch: dict, batch_idx: int = 0) -> torch.Tensor:
        """Reference :46-54 (there the value is also logged): the score-matching loss WITH its tape, for ``loss.backward()`` and an
        optimiser step.  The score network's parameters are created frozen (the sampling path never differentiates); training needs
        ``module.Score.score_net.requires_grad_(True)`` first, on the GPU."""
        if not getattr(self.Score.score_net, "trainable", False):
            raise RuntimeError("training_step: the score network's parameters are frozen - call Score.score_net.requires_grad_(True) "
                               "(validation_step / test_step give the loss without a tape)")
        with torch.enable_grad():
            loss = self.get_score_loss(batch)
        self._log("train/loss_Score", loss, on_step=True, on_epoch=True, prog_bar=True)
        if HAS_LIGHTNING and getattr(self, "_trainer", None) is not None:       # reference :51-53: the learning rate, per epoch
            self._log("lr", self.optimizers().param_groups[0]["lr"], on_step=False, on_epoch=True, prog_bar=True)
        return loss

    def configure_optimizers(self):
        """Reference :26-40: ``optimizer(params=Score.parameters())`` and ``scheduler(optimizer=...)`` from the constructor's partials,
        in Lightning's dictionary layout (the monitor key is the reference's)."""
        if self.optimizer is None:
            raise RuntimeError("configure_optimizers: SGMSEModule was built without an optimizer factory")
        opt = self.optimizer(params=self.Score.parameters())
        if self.scheduler is None:
            return [{"optimizer": opt}]
        return [{"optimizer": opt,
                 "lr_scheduler": {"scheduler": self.scheduler(optimizer=opt), "monitor": "val/loss_Score_epoch", "interval": "epoch",
                                  "frequency": 1}}]
