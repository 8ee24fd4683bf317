"""Predict entry point with the reference's command-line shape (``src/predict.py:39-92``; README.md:176):

    python -m universal_speech_enhancement_amd.predict model=SGMSE_Large ckpt_path=last.ckpt \
        data.data_folder=noisy/ data.target_folder=enhanced/ [model.Score.precision=fp32] [model.sampler_kwargs.N=30]

Hydra and Lightning are not available on the target image, so this is a small stand-in: the same YAML groups
(``configs/predict.yaml`` -> ``data/``, ``model/``), ``key=value`` / ``group=name`` overrides, ``_target_`` instantiation,
and a loop that plays the part of ``trainer.predict`` (one process per GPU under ``torch.distributed.run``)."""
from __future__ import annotations

import importlib
import os
import re
import sys

import torch
import yaml

from . import distributed as D

CONFIG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs")


_FLOAT = re.compile(r"^[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?$")


def _scalars(node):
    """PyYAML (YAML 1.1) reads ``3e-2`` as a string; OmegaConf reads a float -- follow OmegaConf."""
    if isinstance(node, dict):
        return {k: _scalars(v) for k, v in node.items()}
    if isinstance(node, list):
        return [_scalars(v) for v in node]
    if isinstance(node, str) and _FLOAT.match(node):
        return float(node)
    return node


def _load_yaml(*parts):
    with open(os.path.join(CONFIG_DIR, *parts)) as f:
        return _scalars(yaml.safe_load(f) or {})


def _set(cfg: dict, dotted: str, value):
    node = cfg
    keys = dotted.split(".")
    for k in keys[:-1]:
        node = node.setdefault(k, {})
    node[keys[-1]] = value


def compose(overrides) -> dict:
    """defaults list + group selection (``model=NAME``) + dotted overrides (values parsed as YAML scalars)."""
    root = _load_yaml("predict.yaml")
    groups = {}
    for d in root.pop("defaults", []):
        groups.update(d)
    dotted = []
    for ov in overrides:
        k, _, v = ov.partition("=")
        if k in groups and "." not in k:
            groups[k] = v
        else:
            dotted.append((k, _scalars(yaml.safe_load(v))))
    cfg = dict(root)
    for g, name in groups.items():
        cfg[g] = _load_yaml(g, f"{name}.yaml")
    for k, v in dotted:
        _set(cfg, k, v)
    return cfg


def instantiate(node, **extra):
    """Minimal ``hydra.utils.instantiate``: dicts with ``_target_`` become objects, recursively."""
    if isinstance(node, dict):
        kw = {k: instantiate(v) for k, v in node.items() if k != "_target_"}
        if "_target_" in node:
            mod, _, name = node["_target_"].rpartition(".")
            return getattr(importlib.import_module(mod), name)(**kw, **extra)
        return kw
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    return node


def predict(cfg: dict):
    rank, world, local = D.init_from_env()
    torch.cuda.set_device(local)
    data = instantiate(cfg["data"], rank=rank, world_size=world)
    model = instantiate(cfg["model"])
    if cfg.get("ckpt_path") and str(cfg["ckpt_path"]).endswith(".usehip"):   # packed weight file (pack_checkpoint)
        owner = model.G if hasattr(model, "G") else model.Score            # the module that carries n_fft
        net = model.G.net if hasattr(model, "G") else model.Score.score_net
        net.load_weight_file(cfg["ckpt_path"], n_freq=int(owner.n_fft) // 2 + 1, device=torch.device("cuda", local))
    elif cfg.get("ckpt_path"):
        model.load_lightning_checkpoint(cfg["ckpt_path"])
    elif cfg.get("random_init_seed") is not None:
        from .testing.weights import LARGE, REFINE, make_state_dict
        refine = hasattr(model, "G")                          # model=LSGAN: the refine-stage generator
        sd = make_state_dict(int(cfg["random_init_seed"]), **(REFINE if refine else LARGE))
        (model.G.net if refine else model.Score.score_net).load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    else:
        raise SystemExit("ckpt_path is required (or random_init_seed=<int> for a dry run)")
    n = 0
    with torch.no_grad():
        for i, batch in enumerate(data.predict_batches(device=torch.device("cuda", local))):
            model.predict_step(batch, i)
            n += len(batch["name"])
    print(f"[rank {rank}] enhanced {n} file(s) -> {cfg['data']['target_folder']}")
    return n


def main(argv=None):
    predict(compose(list(sys.argv[1:] if argv is None else argv)))


if __name__ == "__main__":
    main()
