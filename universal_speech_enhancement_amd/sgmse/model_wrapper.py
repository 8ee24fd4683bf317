"""``ScoreModel`` with the constructor keys, methods and batch-dict contract of the reference's
``sgmse/model_wrapper.py:23-329`` -- STFT glue in core torch (as the reference: ``torch.stft`` :116-122), the
score network and the whole reverse-SDE loop in libuse_hip.so.

Additions over the reference (all optional, defaults reproduce stock behaviour): ``precision`` ("bf16" | "fp32"),
``use_graph``, and ``noise`` / ``seed`` keywords on ``sample`` / ``enhance`` for reproducible runs.
``train_step`` returns the loss with its tape once ``score_net.requires_grad_(True)`` was called (fp32 HIP operators forward and
backward: ``training.py``); with frozen parameters it is the forward-only value of the sampling engine.
"""
from __future__ import annotations

from math import ceil

import torch
import torch.nn as nn

from . import sampling
from .backbones import BackboneRegistry
from .sdes import SDERegistry
from .util.spectral import SpectralGlue, get_window  # noqa: F401  (get_window: part of the reference module's surface)


class ScoreModel(SpectralGlue, nn.Module):
    supports_fused_sampler = True

    def __init__(self, backbone: str = "ncsnpp", sde: str = "ouve", t_eps: float = 3e-2, mode="regen-joint-training",
                 condition="both", loss_type: str = "mse", n_fft=510, hop_length=128, num_frames=256, window="hann",
                 spec_factor=0.15, spec_abs_exponent=0.5, sde_input="denoised", predictor="reverse_diffusion",
                 corrector="none", precision="bf16", use_graph=True):
        super().__init__()
        input_channels = 6 if condition == "both" else 4
        self.score_net = (BackboneRegistry.get_by_name(backbone)(input_channels=input_channels, precision=precision)
                          if backbone != "none" else None)
        self.sde = SDERegistry.get_by_name(sde)()
        self.t_eps, self.condition, self.mode, self.loss_type = t_eps, condition, mode, loss_type
        self._init_spectral(n_fft, hop_length, num_frames, window, spec_factor, spec_abs_exponent)
        self.sde_input, self.predictor, self.corrector = sde_input, predictor, corrector
        self.precision, self.use_graph = precision, use_graph

    # STFT glue (reference :92-122): SpectralGlue

    # ---- score function (reference :135-145) -------------------------------------------------------------
    def forward_score(self, x, t, score_conditioning, sde_input):
        dnn_input = torch.cat([x] + list(score_conditioning), dim=1)
        return -self.score_net(dnn_input, t)

    def forward(self, x, t, score_conditioning, sde_input):
        return self.forward_score(x, t, score_conditioning, sde_input)

    def _loss(self, err):
        """Reference :124-133."""
        if self.loss_type == "mse":
            losses = torch.square(err.abs())
        elif self.loss_type == "mae":
            losses = err.abs()
        else:
            raise NotImplementedError(f"loss_type {self.loss_type!r}")
        return torch.mean(0.5 * torch.sum(losses.reshape(losses.shape[0], -1), dim=-1))

    def train_step(self, batch, t=None, z=None, start=None):
        """The denoising-score-matching loss of one batch (reference :147-208): crop / pad to ``target_len``, spectrograms,
        t ~ U(t_eps, T), x_t = mean(x0, t, y) + std(t) z, err = score(x_t) std + z, ``_loss(err)``.
        With gradients enabled and a trainable score network (``score_net.requires_grad_(True)``) the network runs on the
        differentiable fp32 HIP operators (``training.ncsnpp_forward_train``) and the loss carries the tape -- what
        ``SGMSEModule.training_step`` returns (SGMSE_module.py:46-54).  Otherwise (``validation_step`` / ``test_step``,
        SGMSE_module.py:56-63, or frozen parameters) it is the forward-only value from the sampling engine.
        ``t`` [B], ``z`` complex [B,1,F,T] and ``start`` override the random draws (the reference draws them from the global
        numpy / torch generators)."""
        if torch.is_grad_enabled() and getattr(self.score_net, "trainable", False):
            return self._train_step(batch, t, z, start)
        with torch.no_grad():
            return self._train_step(batch, t, z, start)

    def _train_step(self, batch, t, z, start):
        import numpy as np
        import torch.nn.functional as F
        x, y = batch["clean"], batch["perturbed"]
        y_denoised = batch.get("fake")
        # The reference feeds spec_fwd(stft(.)) of the target_len excerpt - exactly num_frames frames, NOT padded - to the network, whose
        # len(ch_mult) - 1 FIR down / up sampling stages only close for multiples of 2^(levels - 1) frames (64 for the 7-level NCSN++).  _spectrogram() would zero-pad another count silently
        # (extra frames in z, in the loss and in the network input): refuse it, as the reference's own forward pass fails there.
        ch_mult = getattr(self.score_net, "ch_mult", None)              # backbones without resampling stages take any frame count
        mult = 1 << (len(ch_mult) - 1) if ch_mult else 1
        if self.num_frames % mult != 0:
            raise ValueError(f"train_step: num_frames={self.num_frames} is not a multiple of {mult} frames (the excerpt's spectrogram must "
                             "enter the network unpadded, reference model_wrapper.py:168-171)")
        current_len = x.size(-1)
        pad = max(self.target_len - current_len, 0)
        if pad == 0:                                                     # a random target_len excerpt
            if start is None:
                start = int(np.random.uniform(0, current_len - self.target_len))
            cut = lambda a: a[..., start:start + self.target_len]       # noqa: E731
        else:                                                            # centre the short utterance in zeros
            cut = lambda a: F.pad(a, (pad // 2, pad // 2 + (pad % 2)), mode="constant")   # noqa: E731
        X, Y = self._spectrogram(cut(x).contiguous()), self._spectrogram(cut(y).contiguous())
        Yd = None if y_denoised is None else self._spectrogram(cut(y_denoised).contiguous())
        if self.sde_input == "denoised" and Yd is not None:
            sde_input = Yd
        elif self.sde_input == "noisy":
            sde_input = Y
        else:
            raise NotImplementedError(f"Don't know the sde input you have wished for: {self.sde_input}")
        if t is None:
            t = torch.rand(X.shape[0], device=X.device) * (self.sde.T - self.t_eps) + self.t_eps
        t = t.to(device=X.device, dtype=torch.float32)
        mean, std = self.sde.marginal_prob(X, t, sde_input)
        if z is None:
            z = torch.randn_like(X)                                      # complex: variance 1/2 per component
        sigmas = std.view(-1, 1, 1, 1)
        perturbed = mean + sigmas * z
        if self.condition == "noisy":
            score_conditioning = [Y]
        elif self.condition == "denoised" and Yd is not None:
            score_conditioning = [Yd]
        elif self.condition == "both" and Yd is not None:
            score_conditioning = [Y, Yd]
        else:
            raise NotImplementedError(f"Don't know the conditioning you have wished for: {self.condition}")
        score = self.forward_score(perturbed, t, score_conditioning, sde_input)
        return self._loss(score * sigmas + z)

    # ---- samplers (reference :210-260) -------------------------------------------------------------------
    def fused_sample(self, y, N, predictor, corrector, corrector_steps, snr, t_eps, noise=None, seed=0, use_graph=True,
                     sde=None, cond=None, cond2=None):
        """Whole PC loop inside libuse_hip.so (``use_sample_cond2``), with the OUVE constants of ``sde`` (default: ``self.sde``);
        ``cond``: the score conditioning when it is not ``y`` itself; ``cond2``: the second one of condition="both"."""
        sde = self.sde if sde is None else sde
        eng = self.score_net.engine(y.shape[2], y.device, sde_constants=(sde.theta, sde.sigma_min, sde.sigma_max))
        eng.plan(y.shape[0], y.shape[3])
        eng.set_sampler(N, predictor, corrector, corrector_steps, snr, t_eps, use_graph=use_graph)
        return eng.sample(y, noise=noise, seed=seed, cond=cond, cond2=cond2)

    def get_pc_sampler(self, predictor_name, corrector_name, y, N=None, minibatch=None, **kwargs):
        N = self.sde.N if N is None else N
        sde = self.sde.copy()
        sde.N = N
        kwargs = {"eps": self.t_eps, "use_graph": self.use_graph, **kwargs}
        if minibatch is None:
            return sampling.get_pc_sampler(predictor_name, corrector_name, sde=sde, score_fn=self, y=y, **kwargs)
        M = y.shape[0]
        cond = kwargs.pop("conditioning", None)

        def batched_sampling_fn():
            samples, ns = [], []
            for i in range(int(ceil(M / minibatch))):
                y_mini = y[i * minibatch:(i + 1) * minibatch]
                c_mini = None if cond is None else [y_mini if c is y else c[i * minibatch:(i + 1) * minibatch] for c in cond]
                sample, n = sampling.get_pc_sampler(predictor_name, corrector_name, sde=sde, score_fn=self, y=y_mini,
                                                    conditioning=c_mini, **kwargs)()
                samples.append(sample); ns.append(n)
            return torch.cat(samples, dim=0), ns
        return batched_sampling_fn

    def get_ode_sampler(self, *args, **kwargs):
        raise NotImplementedError("the probability-flow ODE sampler (scipy RK45 host solver) is not on the predict path")

    def sample(self, batch, sampler_type="pc", N=50, corrector_steps=1, snr=0.5, noise=None, seed=0):
        """Reference :262-329: adds ``batch['enhanced']`` (float32 [B, L]) for condition / sde_input 'noisy'."""
        y = batch["perturbed"]
        T_orig = y.size(1)
        Y = self._spectrogram(y)
        Y_denoised = self._spectrogram(batch["fake"]) if "fake" in batch else None
        # conditioning (reference :283-291): the spectrogram(s) the network sees beside x
        if self.condition == "noisy":
            score_conditioning = [Y]
        elif self.condition == "denoised" and Y_denoised is not None:
            score_conditioning = [Y_denoised]
        elif self.condition == "both" and Y_denoised is not None:
            score_conditioning = [Y, Y_denoised]
        else:
            raise NotImplementedError(f"Don't know the conditioning you have wished for: {self.condition}")
        # the SDE's y (reference :293-300)
        if self.sde_input == "denoised" and Y_denoised is not None:
            sde_input = Y_denoised
        elif self.sde_input == "noisy":
            sde_input = Y
        else:
            raise NotImplementedError(f"Don't know the sde input you have wished for: {self.sde_input}")
        if sampler_type != "pc":
            raise NotImplementedError(f"{sampler_type} is not a valid sampler type!")
        sampler = self.get_pc_sampler(self.predictor, self.corrector, sde_input, N=N, corrector_steps=corrector_steps, snr=snr,
                                      intermediate=False, conditioning=score_conditioning, noise=noise, seed=seed)
        sample, nfe = sampler()
        # reference :320-328: the key depends on what the SDE started from
        batch["fake_sde_enhanced" if (self.sde_input == "denoised" and Y_denoised is not None) else "enhanced"] = self._waveform(sample, T_orig)
        return batch

    @torch.no_grad()
    def enhance(self, y, sampler_type="pc", predictor="reverse_diffusion", corrector="ald", N=50, corrector_steps=1,
                snr=0.5, timeit=False, return_stft=False, noise=None, seed=0, sr=24000, **kwargs):
        """One-call enhancement of noisy speech ``y`` [1, L] -- keyword surface of the legacy
        ``ScoreModel.enhance`` (reference ``sgmse/model.py:351-402``)."""
        import time
        start = time.time()
        T_orig = y.size(1)
        norm_factor = y.abs().max().item()
        y = y / norm_factor
        if not y.is_cuda:
            y = y.cuda()
        Y = self._spectrogram(y)
        if sampler_type != "pc":
            raise NotImplementedError(f"{sampler_type} is not a valid sampler type!")
        sample, nfe = self.get_pc_sampler(predictor, corrector, Y, N=N, corrector_steps=corrector_steps, snr=snr,
                                          intermediate=False, conditioning=[Y], noise=noise, seed=seed)()
        if return_stft:
            return sample.squeeze(), Y.squeeze(), T_orig, norm_factor
        x_hat = (self._waveform(sample, T_orig) * norm_factor).squeeze().cpu()
        if timeit:
            return x_hat, nfe, (time.time() - start) / (len(x_hat) / sr)
        return x_hat
