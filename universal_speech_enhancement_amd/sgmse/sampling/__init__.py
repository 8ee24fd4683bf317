"""Predictor-corrector sampling with the reference's call surface
(``sgmse/sampling/__init__.py:23-73`` ``get_pc_sampler``).

Two execution paths, same numerics:
  * fused  -- when ``score_fn`` is the HIP-backed ``ScoreModel`` and predictor / corrector are built-ins, the whole
              loop (prior sampling, N x (corrector, predictor), all score evaluations) runs inside
              ``use_sample`` and is replayed as one hipGraph;
  * seam   -- any other ``score_fn`` callable or user-registered predictor / corrector: the loop below drives
              ``update_fn`` exactly like the reference (corrector before predictor, returns ``x_mean``).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

from .correctors import Corrector, CorrectorRegistry, _HipCorrector, NoneCorrector
from .predictors import Predictor, PredictorRegistry, _HipPredictor, NonePredictor

__all__ = ["PredictorRegistry", "CorrectorRegistry", "Predictor", "Corrector", "get_pc_sampler"]

_SDE_ENGINES: Dict[Tuple, object] = {}


def _sde_engine(sde, device):
    """A weight-less ``use_handle`` carrying the SDE constants, for the stand-alone ``use_sde_*`` kernels."""
    from ...hip_engine import HipScoreEngine
    dev = torch.device(device).index
    dev = torch.cuda.current_device() if dev is None else dev
    key = (dev, float(sde.theta), float(sde.sigma_min), float(sde.sigma_max))
    if key not in _SDE_ENGINES:
        _SDE_ENGINES[key] = HipScoreEngine(device=dev, theta=sde.theta, sigma_min=sde.sigma_min, sigma_max=sde.sigma_max)
    return _SDE_ENGINES[key]


def get_pc_sampler(predictor_name, corrector_name, sde, score_fn, y, denoise=True, eps=3e-2, snr=0.1,
                   corrector_steps=1, probability_flow: bool = False, conditioning=None, intermediate=False,
                   noise=None, seed=0, use_graph=True, **kwargs):
    """Returns ``pc_sampler() -> (x_result, nfe)``.

    Extra keyword arguments over the reference: ``noise`` (complex64 [n_draws, *y.shape], consumed prior-first
    then per step corrector draws followed by the predictor draw) for bit-reproducible parity runs, ``seed`` for
    the device Philox generator, ``use_graph``.
    """
    predictor_cls = PredictorRegistry.get_by_name(predictor_name)
    corrector_cls = CorrectorRegistry.get_by_name(corrector_name)
    predictor = predictor_cls(sde, score_fn, probability_flow=probability_flow)
    corrector = corrector_cls(sde, score_fn, snr=snr, n_steps=corrector_steps)

    builtin = (isinstance(predictor, (_HipPredictor, NonePredictor)) and isinstance(corrector, (_HipCorrector, NoneCorrector))
               and type(predictor) in (PredictorRegistry.get_by_name(n) for n in ("reverse_diffusion", "euler_maruyama", "none"))
               and type(corrector) in (CorrectorRegistry.get_by_name(n) for n in ("langevin", "ald", "none")))
    from ..sdes import OUVESDE
    # the fused loop implements the OUVE dynamics of exactly this class (subclasses / other registered SDEs take the seam path);
    # its constants travel with the call so that the engine is rebuilt when they differ from the defaults
    fused = (builtin and type(sde) is OUVESDE and not probability_flow and denoise
             and getattr(score_fn, "supports_fused_sampler", False)
             and conditioning is not None and len(conditioning) in (1, 2) and all(c.shape == y.shape for c in conditioning))

    if fused:
        def pc_sampler():
            with torch.no_grad():
                x = score_fn.fused_sample(y, N=sde.N, predictor=predictor_name, corrector=corrector_name,
                                          corrector_steps=corrector_steps, snr=snr, t_eps=eps, noise=noise, seed=seed,
                                          use_graph=use_graph, sde=sde, cond=conditioning[0],
                                          cond2=conditioning[1] if len(conditioning) == 2 else None)
            return x, sde.N * (corrector.n_steps + 1)
        return pc_sampler

    def pc_sampler():
        with torch.no_grad():
            draw = 0

            def nz(k=1):
                nonlocal draw
                if noise is None:
                    draw += k
                    return None
                out = [noise[draw + i] for i in range(k)]
                draw += k
                return out

            z0 = nz()
            xt = sde.prior_sampling(y.shape, y, noise=None if z0 is None else z0[0], seed=seed)
            timesteps = torch.linspace(sde.T, eps, sde.N, device=y.device)
            xt_mean = xt
            for i in range(sde.N):
                vec_t = torch.ones(y.shape[0], device=y.device) * timesteps[i]
                if corrector.n_steps:
                    xt, xt_mean = corrector.update_fn(xt, vec_t, y, conditioning=conditioning, noise=nz(corrector.n_steps),
                                                      seed=seed + 1000 * (i + 1)) if isinstance(corrector, _HipCorrector) \
                        else corrector.update_fn(xt, vec_t, y, conditioning=conditioning)
                if isinstance(predictor, _HipPredictor):
                    zp = nz()
                    xt, xt_mean = predictor.update_fn(xt, vec_t, y, conditioning=conditioning,
                                                      noise=None if zp is None else zp[0], seed=seed + 1000 * (i + 1) + 999)
                else:
                    xt, xt_mean = predictor.update_fn(xt, vec_t, y, conditioning=conditioning)
            x_result = xt_mean if (denoise and sde.N) else xt
            return x_result, sde.N * (corrector.n_steps + 1)

    return pc_sampler
