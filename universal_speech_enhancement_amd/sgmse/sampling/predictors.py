"""Predictor registry and built-in predictors with the reference's interface
(``sgmse/sampling/predictors.py``: ``PredictorRegistry`` :8, ``Predictor`` :11-37, ``euler_maruyama`` :40-53,
``reverse_diffusion`` :56-68, ``none`` :71-79).  The built-ins evaluate ``score_fn`` and then run the
element-wise update in libuse_hip.so (``use_sde_predictor``)."""
from __future__ import annotations

import abc

import torch

from ..util.registry import Registry

PredictorRegistry = Registry("Predictor")


class Predictor(abc.ABC):
    def __init__(self, sde, score_fn, probability_flow=False):
        super().__init__()
        self.sde = sde
        self.rsde = sde.reverse(score_fn) if hasattr(sde, "reverse") else None       # (sic: the reference drops probability_flow here, predictors.py:17)
        self.score_fn = score_fn
        self.probability_flow = probability_flow

    @abc.abstractmethod
    def update_fn(self, x, t, *args, **kwargs):
        """One predictor step: returns (x_next, x_mean)."""

    def debug_update_fn(self, x, t, *args):
        raise NotImplementedError(f"Debug update function not implemented for predictor {self}.")


def _eval_score(score_fn, x, t, args, kwargs):
    cond = kwargs.get("conditioning")
    if cond is not None:
        return score_fn(x, t, score_conditioning=cond, sde_input=args[0])
    return score_fn(x, t, *args)


def _uniform_t(t: torch.Tensor) -> float:
    tv = t.reshape(-1)
    t0 = float(tv[0])
    if tv.numel() > 1 and not bool((tv == tv[0]).all()):
        raise ValueError("the device update kernels take one time value per call; split the batch for per-item t")
    return t0


class _HipPredictor(Predictor):
    hip_name = ""

    def update_fn(self, x, t, *args, noise=None, seed=0, **kwargs):
        from . import _sde_engine
        y = args[0]
        score = _eval_score(self.score_fn, x, t, args, kwargs)
        # (probability_flow: stored and, exactly as in the reference, without effect on the update - Predictor.__init__ builds its
        # reverse SDE as sde.reverse(score_fn), predictors.py:17; pinned by tests/golden/sampler_pf_*.npz)
        return _sde_engine(self.sde, x.device).sde_predictor(self.hip_name, _uniform_t(t), self.sde.N, x, y, score,
                                                             noise=noise, seed=seed)


@PredictorRegistry.register("euler_maruyama")
class EulerMaruyamaPredictor(_HipPredictor):
    hip_name = "euler_maruyama"


@PredictorRegistry.register("reverse_diffusion")
class ReverseDiffusionPredictor(_HipPredictor):
    hip_name = "reverse_diffusion"


@PredictorRegistry.register("none")
class NonePredictor(Predictor):
    """Does nothing."""

    def __init__(self, *args, **kwargs):
        pass

    def update_fn(self, x, t, *args, **kwargs):
        return x, x
