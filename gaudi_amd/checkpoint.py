"""Reference checkpoint format: ``<exp_dir>/args.txt`` (json of the argparse namespace) + ``<exp_dir>/model.pt``
(``torch.save(state_dict)``), keys optionally prefixed ``module.`` (models_edm.py:98-102).
Mirrors utils/helpers.py:204-224 (get_edm_args / get_cond_predictor_args)."""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import numpy as np


def _load_args(exp_dir_path: str) -> SimpleNamespace:
    with open(os.path.join(exp_dir_path, "args.txt"), "r") as f:
        d = json.load(f)
    a = SimpleNamespace(**d)
    a.restore = True
    a.exp_dir = exp_dir_path
    a.device = "cuda"  # the reference picks cuda if available; this framework has no CPU path
    return a


def get_edm_args(exp_dir_path: str) -> SimpleNamespace:
    """utils/helpers.py:204-213."""
    return _load_args(exp_dir_path)


def get_cond_predictor_args(exp_dir_path: str) -> SimpleNamespace:
    """utils/helpers.py:215-224."""
    return _load_args(exp_dir_path)


def load_state_dict(exp_dir_path: str) -> dict:
    """model.pt -> {name: float32 ndarray} with the ``module.`` prefix stripped.  torch.load on the host is
    fine here: it is outside the sampling loop."""
    import torch

    sd = torch.load(os.path.join(exp_dir_path, "model.pt"), map_location="cpu")
    out = {}
    for k, v in sd.items():
        out[k[7:] if k.startswith("module.") else k] = np.ascontiguousarray(v.detach().cpu().numpy().astype(np.float32))
    return out


def args_dict(args) -> dict:
    return dict(vars(args)) if not isinstance(args, dict) else dict(args)


def normalize_factors(args: dict):
    """args.normalize_factors; a namespace without the key gets the reference's argparse default [3, 4, 10]
    (utils/args_edm.py:48), never an identity normalisation."""
    nv = args.get("normalize_factors")
    if nv is None:
        nv = [3, 4, 10]
    nv = [float(v) for v in nv]
    if len(nv) != 3 or not all(v > 0 for v in nv):
        raise ValueError(f"normalize_factors must be three positive numbers, got {nv}")
    return nv
