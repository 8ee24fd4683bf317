"""SDE registry and the OUVE SDE with the interface of the reference's ``sgmse/sdes.py``
(``SDERegistry`` :17, ``SDE`` ABC :20-175, ``OUVESDE`` :182-279).

The heavy lifting of the predict path does not go through these tensor methods: the built-in
predictors / correctors and the fused sampler call libuse_hip.so.  They exist so that user-registered
predictors / correctors written against the reference API (``sde.sde``, ``sde.discretize``,
``sde.reverse(score_fn)``, ``sde.marginal_prob`` ...) keep working on CUDA tensors.
"""
from __future__ import annotations

import abc
import warnings

import numpy as np
import torch

from .util.registry import Registry

SDERegistry = Registry("SDE")


class SDE(abc.ABC):
    """Forward SDE dx = f(x, t) dt + g(t) dw on mini-batches (reference sdes.py:20-92)."""

    def __init__(self, N: int):
        super().__init__()
        self.N = N

    @property
    @abc.abstractmethod
    def T(self):
        ...

    @abc.abstractmethod
    def sde(self, x, t, *args):
        ...

    @abc.abstractmethod
    def marginal_prob(self, x, t, *args):
        ...

    @abc.abstractmethod
    def prior_sampling(self, shape, *args):
        ...

    @abc.abstractmethod
    def copy(self):
        ...

    def discretize(self, x, t, *args):
        """x_{i+1} = x_i + f_i + G_i z_i with the Euler-Maruyama rule (reference sdes.py:75-92)."""
        dt = 1 / self.N
        drift, diffusion = self.sde(x, t, *args)
        return drift * dt, diffusion * torch.sqrt(torch.tensor(dt, device=t.device))

    def reverse(self, score_model, probability_flow: bool = False):
        """Reverse-time SDE / probability-flow ODE bound to ``score_model`` (reference sdes.py:94-175)."""
        fwd = self
        pf = 0.5 if probability_flow else 1.0

        class RSDE:
            N = fwd.N
            T = fwd.T

            def __init__(self):
                self.probability_flow = probability_flow

            @staticmethod
            def _score(x, t, args, kwargs):
                cond = kwargs.get("conditioning")
                if cond is not None:
                    return score_model(x, t, score_conditioning=cond, sde_input=args[0])
                return score_model(x, t, *args)

            def sde(self, x, t, *args, **kwargs):
                drift, diffusion = fwd.sde(x, t, *args)
                if diffusion.ndim < x.ndim:
                    diffusion = diffusion.view(*diffusion.size(), *((1,) * (x.ndim - diffusion.ndim)))
                total = drift - diffusion ** 2 * self._score(x, t, args, kwargs) * pf
                return total, (torch.zeros_like(diffusion) if probability_flow else diffusion)

            def discretize(self, x, t, *args, **kwargs):
                f, G = fwd.discretize(x, t, *args)
                if G.ndim < x.ndim:
                    G = G.view(*G.size(), *((1,) * (x.ndim - G.ndim)))
                rev_f = f - G ** 2 * self._score(x, t, args, kwargs) * pf
                return rev_f, (torch.zeros_like(G) if probability_flow else G)

        return RSDE()


@SDERegistry.register("ouve")
class OUVESDE(SDE):
    """dx = theta (y - x) dt + sigma(t) dw,  sigma(t) = sigma_min (sigma_max/sigma_min)^t sqrt(2 log(sigma_max/sigma_min))
    (reference sdes.py:182-254)."""

    def __init__(self, theta=1.5, sigma_min=0.05, sigma_max=0.5, N=1000, **ignored_kwargs):
        super().__init__(N)
        self.theta, self.sigma_min, self.sigma_max = theta, sigma_min, sigma_max
        self.logsig = np.log(self.sigma_max / self.sigma_min)

    def copy(self):
        return OUVESDE(self.theta, self.sigma_min, self.sigma_max, N=self.N)

    @property
    def T(self):
        return 1

    def sde(self, x, t, y):
        sigma = self.sigma_min * (self.sigma_max / self.sigma_min) ** t
        return self.theta * (y - x), sigma * np.sqrt(2 * self.logsig)

    def _mean(self, x0, t, y):
        w = torch.exp(-self.theta * t)[:, None, None, None]
        return w * x0 + (1 - w) * y

    def _std(self, t, **kwargs):
        th, ls, sm = self.theta, self.logsig, self.sigma_min
        return torch.sqrt(sm ** 2 * torch.exp(-2 * th * t) * (torch.exp(2 * (th + ls) * t) - 1) * ls / (th + ls))

    def marginal_prob(self, x0, t, y):
        return self._mean(x0, t, y), self._std(t)

    def prior_sampling(self, shape, y, noise=None, seed=0):
        """x_T = y + z std(1).  Runs on the device through ``use_sde_prior`` (reference sdes.py:248-254)."""
        if tuple(shape) != tuple(y.shape):
            warnings.warn(f"Target shape {shape} does not match shape of y {y.shape}! Ignoring target shape.")
        from ..hip_engine import UseHipError
        from .sampling import _sde_engine
        if not y.is_cuda:
            raise UseHipError("prior_sampling needs a CUDA (ROCm) tensor; there is no CPU sampling path")
        return _sde_engine(self, y.device).sde_prior(y, noise=noise, seed=seed)

    def prior_logp(self, z):
        raise NotImplementedError("prior_logp for OU SDE not yet implemented!")
