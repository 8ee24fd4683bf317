"""Corrector registry and built-in correctors with the reference's interface
(``sgmse/sampling/correctors.py``: ``CorrectorRegistry`` :8, ``Corrector`` :11-34, ``langevin`` :37-63,
``ald`` :66-98, ``none`` :101-111).  The built-ins evaluate ``score_fn`` and run the norm reductions and the
element-wise update in libuse_hip.so (``use_sde_corrector``)."""
from __future__ import annotations

import abc

from ..util.registry import Registry
from .predictors import _eval_score, _uniform_t

CorrectorRegistry = Registry("Corrector")


class Corrector(abc.ABC):
    def __init__(self, sde, score_fn, snr, n_steps):
        super().__init__()
        self.sde = sde
        self.rsde = sde.reverse(score_fn) if hasattr(sde, "reverse") else None
        self.score_fn = score_fn
        self.snr = snr
        self.n_steps = n_steps

    @abc.abstractmethod
    def update_fn(self, x, t, *args, **kwargs):
        """One corrector call (n_steps inner steps): returns (x_next, x_mean)."""


class _HipCorrector(Corrector):
    hip_name = ""

    def update_fn(self, x, t, *args, noise=None, seed=0, **kwargs):
        """``noise``: optional list of n_steps complex64 tensors (consumed in order)."""
        from . import _sde_engine
        eng = _sde_engine(self.sde, x.device)
        x_mean = x
        for k in range(self.n_steps):
            grad = _eval_score(self.score_fn, x, t, args, kwargs)
            z = None if noise is None else noise[k]
            x, x_mean = eng.sde_corrector(self.hip_name, _uniform_t(t), self.snr, x, grad, noise=z, seed=seed + k)
        return x, x_mean


@CorrectorRegistry.register(name="langevin")
class LangevinCorrector(_HipCorrector):
    """Step size 2 (snr * mean_b||z_b|| / mean_b||g_b||)^2 -- means over the batch (reference :55-57)."""
    hip_name = "langevin"


@CorrectorRegistry.register(name="ald")
class AnnealedLangevinDynamics(_HipCorrector):
    """Step size 2 (snr * std(t))^2 (reference :79-98); OUVE only."""
    hip_name = "ald"

    def __init__(self, sde, score_fn, snr, n_steps):
        from ..sdes import OUVESDE
        if not isinstance(sde, OUVESDE):
            raise NotImplementedError(f"SDE class {sde.__class__.__name__} not yet supported.")
        super().__init__(sde, score_fn, snr, n_steps)


@CorrectorRegistry.register(name="none")
class NoneCorrector(Corrector):
    """Does nothing."""

    def __init__(self, *args, **kwargs):
        self.snr = 0
        self.n_steps = 0

    def update_fn(self, x, t, *args, **kwargs):
        return x, x
