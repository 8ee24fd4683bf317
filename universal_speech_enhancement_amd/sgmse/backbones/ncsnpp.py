"""HIP-backed NCSN++ score network behind the reference's backbone interface.

``BackboneRegistry.get_by_name("ncsnpplarge")(input_channels=4)`` returns an ``nn.Module`` whose
``state_dict()`` has exactly the reference's keys and shapes (reference ``sgmse/backbones/ncsnpp.py:116-316``:
``all_modules.<i>....``, ``output_layer.*``), so Lightning checkpoints of the reference load unchanged, and whose
``forward(x, time_cond)`` (reference :324-501: x complex [B,2,F,T'] = cat[x_t, Y], time_cond [B] ->
complex [B,1,F,T']) runs in libuse_hip.so.  The module holds parameters only; there is no PyTorch forward.
"""
from __future__ import annotations

from typing import Sequence

import numpy as np
import torch
import torch.nn as nn

from .arch import ncsnpp_param_shapes
from .shared import BackboneRegistry


class _Holder(nn.Module):
    """Parameter container; nesting reproduces the reference's dotted key names."""


def _fan_avg_uniform(shape, scale, gen):
    """``default_init(scale)`` = variance_scaling(scale, 'fan_avg', 'uniform') (reference layers.py:66-103)."""
    scale = 1e-10 if scale == 0 else scale
    rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[1] * rf, shape[0] * rf
    bound = float(np.sqrt(3.0 * scale / ((fan_in + fan_out) / 2.0)))
    return (torch.rand(*shape, generator=gen) * 2.0 - 1.0) * bound


@BackboneRegistry.register("ncsnpp")
class NCSNpp(nn.Module):
    """NCSN++ with biggan res-blocks, FIR resampling, progressive 'output_skip' / 'input_skip' (sum), Fourier
    time embedding and bottleneck attention -- the configuration family of the predict path
    (reference ncsnpp.py:42-69 defaults)."""

    supports_hip = True

    def __init__(self, nf=128, ch_mult: Sequence[int] = (1, 2, 2, 2), num_res_blocks=1, input_channels=4,
                 fourier_scale=16, init_scale=0.0, image_size=256, precision="bf16", n_freq=None,
                 scale_by_sigma=True, conditional=True, discriminative=False, **unsupported):
        super().__init__()
        fixed = dict(nonlinearity="swish", attn_resolutions=(0,), resamp_with_conv=True,
                     fir=True, fir_kernel=[1, 3, 3, 1], skip_rescale=True, resblock_type="biggan",
                     progressive="output_skip", progressive_input="input_skip", progressive_combine="sum",
                     embedding_type="fourier", spatial_channels=1, dropout=0.0, centered=False)
        for k, v in unsupported.items():
            if k in fixed and (list(v) if isinstance(v, (list, tuple)) else v) != (list(fixed[k]) if isinstance(fixed[k], (list, tuple)) else fixed[k]):
                raise NotImplementedError(f"NCSNpp(HIP): option {k}={v!r} is outside the predict-path configuration ({fixed[k]!r})")
        self.discriminative = bool(discriminative)
        if self.discriminative:      # reference ncsnpp.py:86-92: options that make no sense for a discriminative model
            conditional, scale_by_sigma, input_channels = False, False, 2
        if input_channels not in (2, 4, 6):
            raise NotImplementedError("NCSNpp(HIP): input_channels must be 4 (condition='noisy' / 'denoised'), 2 (discriminative) or "
                                      "6 (condition='both')")
        self.conditional, self.scale_by_sigma = bool(conditional), bool(scale_by_sigma)
        self.nf, self.ch_mult, self.num_res_blocks = nf, tuple(ch_mult), num_res_blocks
        self.input_channels, self.precision, self.n_freq = input_channels, precision, n_freq
        gen = torch.Generator().manual_seed(torch.initial_seed() & 0x7FFFFFFF)
        shapes = ncsnpp_param_shapes(nf=nf, ch_mult=ch_mult, num_res_blocks=num_res_blocks, input_channels=input_channels,
                                     conditional=self.conditional)
        self.output_layer = _Holder()
        self.all_modules = nn.ModuleList()
        zero_init = ("Conv_1.weight", "NIN_3.W")          # init_scale=0 layers (reference layerspp.py:74,273)
        for key, shape in shapes.items():
            parts = key.split(".")
            if len(shape) == 1:
                if key == "all_modules.0.W":
                    val = torch.randn(shape, generator=gen) * fourier_scale
                elif parts[-1] == "weight":
                    val = torch.ones(shape)
                else:
                    val = torch.zeros(shape)
            else:
                pyramid_conv = len(parts) == 3 and shape[0] == input_channels and len(shape) == 4 and shape[2] == 3
                scale = init_scale if (key.endswith(zero_init) or pyramid_conv) else (0.1 if ".NIN_" in key else 1.0)
                val = _fan_avg_uniform(shape, scale, gen)
            node = self
            if parts[0] == "all_modules":
                idx = int(parts[1])
                while len(self.all_modules) <= idx:
                    self.all_modules.append(_Holder())
                node = self.all_modules[idx]
                parts = parts[2:]
            for p in parts[:-1]:
                if not hasattr(node, p):
                    node.add_module(p, _Holder())
                node = getattr(node, p)
            node.register_parameter(parts[-1], nn.Parameter(val, requires_grad=False))
        self.train_precision = "fp32"     # "bf16" / "fp16": mixed-precision taped forward (training.ncsnpp_forward_train)
        self._engine = None
        self._engine_sde = self._DEFAULT_SDE
        self._engine_dirty = True
        self._weight_file = None                             # set by load_weight_file: the engine's weights come from it

        def _on_load(module, incompatible):                  # a state dict replaces whatever a weight file provided
            module._engine_dirty, module._weight_file = True, None
        self.register_load_state_dict_post_hook(_on_load)

    # -- engine management --------------------------------------------------------------------------------
    _DEFAULT_SDE = (1.5, 0.05, 0.5)

    def _new_engine(self, n_freq, device, sde_constants):
        from ...hip_engine import HipScoreEngine
        th, smin, smax = sde_constants
        return HipScoreEngine(nf=self.nf, ch_mult=self.ch_mult, num_res_blocks=self.num_res_blocks, n_freq=n_freq,
                              precision=self.precision, device=None if device is None else torch.device(device).index,
                              theta=th, sigma_min=smin, sigma_max=smax, input_channels=self.input_channels,
                              conditional=self.conditional, scale_by_sigma=self.scale_by_sigma)

    def engine(self, n_freq: int, device=None, sde_constants=None):
        """The ``use_handle`` for ``n_freq`` bins (and the OUVE constants of the fused sampler); rebuilt when either
        changes.  Weights come from the packed file when ``load_weight_file`` was used, else from the module's
        parameters."""
        want = tuple(float(v) for v in (sde_constants or self._engine_sde))
        if self._engine is None or self._engine.n_freq != n_freq or self._engine_sde != want:
            self._engine = self._new_engine(n_freq, device, want)
            self._engine_sde = want
            self._engine_dirty = True
        if self._engine_dirty:
            if self._weight_file is not None:
                self._engine.load_weight_blob(self._weight_file)
            else:
                self._engine.load_state_dict(self.state_dict())
            self._engine_dirty = False
        return self._engine

    def load_weight_file(self, path: str, n_freq: int = 512, device=None):
        """Start from a packed weight file (``pack_checkpoint``; ``use_load_weight_blob``) instead of a state dict: the
        module's own parameters are left untouched (their random initialisation) and no longer consulted -- until ``load_state_dict`` /
        ``refresh_weights`` hands the module's parameters back to the engine.  Training from this state is refused
        (``forward_train`` raises): the packed file cannot be unpacked into parameters."""
        self._weight_file = str(path)
        self._engine = None
        self._engine_dirty = True
        self.engine(n_freq, device)

    def refresh_weights(self):
        """Call after modifying parameters in place (``load_state_dict`` is tracked automatically)."""
        self._engine_dirty, self._weight_file = True, None

    def requires_grad_(self, requires_grad: bool = True):
        """``module.requires_grad_(True)`` is the documented way to enable training.  The Gaussian-Fourier projection ``all_modules.0.W``
        stays frozen, as in the reference (``GaussianFourierProjection``: ``nn.Parameter(..., requires_grad=False)``, layerspp.py:35) -
        an optimiser with weight decay would otherwise move the embedding frequencies under data-parallel training."""
        super().requires_grad_(requires_grad)
        if requires_grad and self.conditional and len(self.all_modules) and hasattr(self.all_modules[0], "W"):
            self.all_modules[0].W.requires_grad_(False)
        return self

    @property
    def trainable(self) -> bool:
        """True once ``requires_grad_(True)`` was called on the module (parameters are created frozen: the sampling path never
        differentiates)."""
        return any(p.requires_grad for p in self.parameters())

    def forward_train(self, x: torch.Tensor, time_cond: torch.Tensor = None) -> torch.Tensor:
        """The same network with a tape (operators of libuse_hip.so forward and backward: ``training.ncsnpp_forward_train``; fp32, or
        mixed precision with ``self.train_precision = "bf16"``);
        what ``forward`` runs when gradients are enabled and the parameters are trainable.  The sampling engine's weight copy is marked
        stale: an optimiser step is assumed to follow."""
        from ...training import ncsnpp_forward_train
        P = dict(self.named_parameters())
        if any(not p.is_cuda for p in P.values()):
            from ...hip_engine import UseHipError
            raise UseHipError("NCSN++ (HIP) training needs the parameters on the GPU (module.to('cuda')): there is no CPU implementation")
        if self._weight_file is not None:
            from ...hip_engine import UseHipError
            raise UseHipError(f"the module's weights came from the packed file '{self._weight_file}' (load_weight_file): its parameters still hold "
                              "their random initialisation, so training would silently discard the file - load a state dict "
                              "(load_state_dict / pack_checkpoint's source checkpoint) before fine-tuning")
        self._engine_dirty = True
        cd = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[self.train_precision]
        return ncsnpp_forward_train(P, x, time_cond, self.ch_mult, self.num_res_blocks, conditional=self.conditional,
                                    scale_by_sigma=self.scale_by_sigma, compute_dtype=cd)

    def forward(self, x: torch.Tensor, time_cond: torch.Tensor = None) -> torch.Tensor:
        nin = self.input_channels // 2
        if x.dim() != 4 or x.shape[1] != nin:
            raise ValueError("expected complex input [B, 2, F, T'] = cat([x_t, Y], dim=1)" if nin == 2 else
                             "expected complex input [B, 3, F, T'] = cat([x_t, Y, Y_denoised], dim=1)" if nin == 3 else
                             "expected complex input [B, 1, F, T'] (discriminative network)")
        if not x.is_cuda:
            from ...hip_engine import UseHipError
            raise UseHipError("NCSN++ (HIP) needs CUDA/ROCm tensors: the network has no CPU implementation")
        if self.conditional and time_cond is None:
            raise ValueError("a conditional NCSN++ needs time_cond")
        if torch.is_grad_enabled() and self.trainable:
            return self.forward_train(x, time_cond)
        eng = self.engine(x.shape[2], x.device)
        if nin == 1:
            return eng.forward(x, None, time_cond if (self.conditional or self.scale_by_sigma) else None)
        if nin == 3:                                         # condition="both": forward = -score (use_score2)
            return -eng.score(x[:, 0:1], x[:, 1:2], time_cond, y2=x[:, 2:3])
        return eng.forward(x[:, 0:1], x[:, 1:2], time_cond)


@BackboneRegistry.register("ncsnpplarge")
class NCSNppLarge(NCSNpp):
    """nf=128, ch_mult=(1,1,2,2,2,2,2), 2 res-blocks per level, ~65 M parameters (reference ncsnpp.py:504-518)."""

    def __init__(self, **kwargs):
        super().__init__(nf=128, ch_mult=(1, 1, 2, 2, 2, 2, 2), num_res_blocks=2, **kwargs)


@BackboneRegistry.register("ncsnpp12M")
class NCSNpp12M(NCSNpp):
    """nf=96, ch_mult=(1,2,2,1), one res-block per level, ~12 M parameters (reference ncsnpp.py:527-541)."""

    def __init__(self, **kwargs):
        super().__init__(nf=96, ch_mult=(1, 2, 2, 1), num_res_blocks=1, **kwargs)


@BackboneRegistry.register("ncsnpp6M")
class NCSNpp6M(NCSNpp):
    """nf=96, ch_mult=(1,1,1,1), one res-block per level, ~6 M parameters (reference ncsnpp.py:545-559)."""

    def __init__(self, **kwargs):
        super().__init__(nf=96, ch_mult=(1, 1, 1, 1), num_res_blocks=1, **kwargs)


# explicit names for configs that want to state the implementation
BackboneRegistry.register("ncsnpp_hip")(NCSNpp)
BackboneRegistry.register("ncsnpplarge_hip")(NCSNppLarge)
