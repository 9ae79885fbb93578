"""Seeded synthetic checkpoints in the reference's on-disk format.

No trained ``model.pt`` / ``args.txt`` ships with the reference (README.md:20-23), so the bench,
the smoke test and the parity tests use state dicts drawn from a counter-based generator
(numpy Philox) with the same shapes, names and init ranges as the reference modules:

* ``nn.Linear`` default init: U(+-1/sqrt(fan_in)) for weight and bias;
* coordinate heads ``coord_mlp.4.weight`` (edm/egnn/egnn_new.py:107-108) and
  ``coord_mlp.2.weight`` (edm/egnn_predictor/gcl.py:205-206): xavier_uniform(gain=1e-3), or
  N(0, 1/H) when ``amplify_coord=True`` so that the coordinate branch matters in parity tests.

Key names follow SURVEY.md section 5 ("checkpoint / resume").
"""
from __future__ import annotations

import json
import os

import numpy as np

EDM_DEFAULTS = dict(  # utils/args_edm.py:10-48
    dataset="cata", max_nodes=11, dp=True, n_layers=9, nf=192, tanh=True, attention=True,
    coords_range=4.0, norm_constant=1.0, sin_embedding=False, inv_sublayers=1,
    normalization_factor=1.0, aggregation_method="sum", diffusion_steps=1000,
    diffusion_noise_schedule="polynomial_2", diffusion_noise_precision=1e-5,
    diffusion_loss_type="l2", normalize_factors=[3, 4, 10],
)

PRED_DEFAULTS = dict(  # cond_prediction/prediction_args.py:10-47
    dataset="cata", max_nodes=11, dp=True, n_layers=12, nf=196, tanh=True, attention=True,
    coords_range=4.0, norm_constant=1.0, normalization_factor=1.0,
    target_features="LUMO_eV,GAP_eV,Erel_eV,aIP_eV,aEA_eV",
)


def num_node_features(dataset: str) -> int:
    """data/aromatic_dataloader.py:31-35,148: cata -> 1 ring type, hetro -> 11 ring types + '.'."""
    return 1 if dataset == "cata" else 12


def edm_args(**over) -> dict:
    a = dict(EDM_DEFAULTS)
    a.update(over)
    return a


def pred_args(**over) -> dict:
    a = dict(PRED_DEFAULTS)
    a.update(over)
    return a


class _Gen:
    def __init__(self, seed):
        self.g = np.random.Generator(np.random.Philox(key=int(seed)))

    def uniform(self, shape, bound):
        return ((self.g.random(shape) * 2.0 - 1.0) * bound).astype(np.float32)

    def normal(self, shape, std):
        return (self.g.standard_normal(shape) * std).astype(np.float32)


def _linear(sd, g, name, fan_out, fan_in, bias=True):
    b = 1.0 / np.sqrt(fan_in)
    sd[name + ".weight"] = g.uniform((fan_out, fan_in), b)
    if bias:
        sd[name + ".bias"] = g.uniform((fan_out,), b)


def _coord_head(sd, g, name, H, amplify):
    if amplify:
        sd[name] = g.normal((1, H), 1.0 / np.sqrt(H))
    else:
        sd[name] = g.uniform((1, H), 1e-3 * np.sqrt(6.0 / (H + 1)))


def synth_edm_state_dict(args: dict, in_node_nf: int, seed: int = 0, amplify_coord: bool = False,
                         gamma: np.ndarray | None = None) -> dict:
    """State dict of EnVariationalDiffusion(EGNN_dynamics) -- models_edm.py:67-96."""
    g = _Gen(seed)
    H = args["nf"]
    F1 = in_node_nf + 1  # + time (condition_time=True, models_edm.py:82)
    EF = 24 if args.get("sin_embedding", False) else 2  # edge features: (r, d0), or 2 x 12 sinusoids of them (egnn_new.py:269-273)
    sd = {}
    sd["buffer"] = np.zeros(1, np.float32)
    if gamma is not None:
        sd["gamma.gamma"] = np.asarray(gamma, np.float32)
    p = "dynamics.egnn."
    _linear(sd, g, p + "embedding", H, F1)
    _linear(sd, g, p + "embedding_out", F1, H)
    for l in range(args["n_layers"]):
        for s in range(args["inv_sublayers"]):
            q = f"{p}e_block_{l}.gcl_{s}."
            _linear(sd, g, q + "edge_mlp.0", H, 2 * H + EF)
            _linear(sd, g, q + "edge_mlp.2", H, H)
            _linear(sd, g, q + "node_mlp.0", H, 2 * H)
            _linear(sd, g, q + "node_mlp.2", H, H)
            if args["attention"]:
                _linear(sd, g, q + "att_mlp.0", 1, H)
        q = f"{p}e_block_{l}.gcl_equiv."
        _linear(sd, g, q + "coord_mlp.0", H, 2 * H + EF)
        _linear(sd, g, q + "coord_mlp.2", H, H)
        _coord_head(sd, g, q + "coord_mlp.4.weight", H, amplify_coord)
    return sd


def synth_predictor_state_dict(args: dict, in_nf: int, out_nf: int = 5, seed: int = 1,
                               amplify_coord: bool = False) -> dict:
    """State dict of EGNN_predictor -- cond_prediction/train_cond_predictor.py:183-196."""
    g = _Gen(seed)
    H = args["nf"]
    sd = {}
    p = "egnn."
    _linear(sd, g, p + "embedding", H, in_nf + 1)
    _linear(sd, g, p + "embedding_out", out_nf, H)
    for l in range(args["n_layers"]):
        q = f"{p}gcl_{l}."
        _linear(sd, g, q + "edge_mlp.0", H, 2 * H + 2)
        _linear(sd, g, q + "edge_mlp.2", H, H)
        _linear(sd, g, q + "node_mlp.0", H, 2 * H)
        _linear(sd, g, q + "node_mlp.2", H, H)
        _linear(sd, g, q + "coord_mlp.0", H, H)
        _coord_head(sd, g, q + "coord_mlp.2.weight", H, amplify_coord)
        if args["attention"]:
            _linear(sd, g, q + "att_mlp.0", 1, H)
    return sd


def write_checkpoint(exp_dir: str, args: dict, sd: dict, module_prefix: bool | None = None) -> None:
    """Write ``args.txt`` (json of the namespace, train_edm.py:207-208) and ``model.pt``
    (``torch.save(state_dict)``, train_edm.py:183).  Keys get the ``module.`` prefix iff
    ``args['dp']`` (models_edm.py:98-102) unless overridden."""
    import torch

    os.makedirs(exp_dir, exist_ok=True)
    with open(os.path.join(exp_dir, "args.txt"), "w") as f:
        json.dump(args, f)
    pre = "module." if (args.get("dp", True) if module_prefix is None else module_prefix) else ""
    tsd = {pre + k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    torch.save(tsd, os.path.join(exp_dir, "model.pt"))
