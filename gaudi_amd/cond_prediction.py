"""Host mirror of the predictor-evaluation path: sample_edm_t / compute_loss (cond_prediction/train_cond_predictor.py:47-81)
and val_epoch / the t-sweep of eval_cond_predictor.py:34-113, on top of gaudi_predict_noised (forward noising fused into the
predictor-forward launch).  Evaluation only: there is no backward pass through the weights here (training is out of scope)."""
from __future__ import annotations

import numpy as np

from ._lib import GaudiError
from .models_edm import _like_ref, _to_numpy


def _t_int(t, T, B):
    """gamma(t) is a table lookup at round(t * T) (en_diffusion.py:220-223)."""
    t = np.asarray(_to_numpy(t), np.float64).reshape(-1)
    return np.broadcast_to(np.round(t * T).astype(np.int64), (B,))


def _run(cond_predictor, edm_model, x, h, node_mask, edge_mask, t, noise=None):
    if cond_predictor.engine is not edm_model.engine:
        raise GaudiError("the predictor must be attached to this model (get_cond_predictor_model(..., model=model))")
    x = _to_numpy(x).astype(np.float32)
    B, N = x.shape[0], x.shape[1]
    nm = _to_numpy(node_mask).astype(np.float32).reshape(B, N)
    em = (nm[:, :, None] * nm[:, None, :] * (1 - np.eye(N, dtype=np.float32)) if edge_mask is None
          else _to_numpy(edge_mask).astype(np.float32).reshape(B, N, N))
    seed, off = edm_model.next_stream(B)  # fresh noise on the next call, as successive torch.randn draws would be
    return edm_model.engine.predict_noised(x, _to_numpy(h).astype(np.float32), _t_int(t, edm_model.T, B), nm, em,
                                           seed=seed, sample_offset=off, noise=noise)


def sample_edm_t(x, h, edm_model, t, node_mask, noise=None, cond_predictor=None):
    """cond_prediction/train_cond_predictor.py:47-61 -> z_t [B,N,3+F].  ``noise`` [B,N,3+F]: raw N(0,1) draws to inject
    (parity tests); None -> the device Philox stream keyed by (model.seed, model.sample_offset + b)."""
    cp = cond_predictor if cond_predictor is not None else getattr(edm_model, "cond_predictor", None)
    if cp is None:
        raise GaudiError("sample_edm_t runs inside the predictor kernel: attach a predictor first "
                         "(get_cond_predictor_model(..., model=model))")
    return _like_ref(_run(cp, edm_model, x, h, node_mask, None, t, noise)[0])


def compute_loss(model, x, h, node_mask, edge_mask, target, edm_model, edm_args, t_fix=None, noise=None):
    """cond_prediction/train_cond_predictor.py:64-81 (forward only) -> (l1 loss, |pred - target| [B,K])."""
    import torch
    T = edm_model.T
    B = _to_numpy(x).shape[0]
    if t_fix is None:
        t_int = torch.randint(0, T + 1, size=(B, 1)).float()  # the reference's draw (torch RNG stream)
    else:
        t_int = torch.ones(B, 1).float() * float(t_fix)
    zt, pred = _run(model, edm_model, x, h, node_mask, edge_mask, t_int / T, noise)
    err = np.abs(pred - _to_numpy(target).astype(np.float32))
    return _like_ref(np.float32(err.mean())), _like_ref(err)


def val_epoch(tag, cond_predictor, edm_model, dataloader, args, edm_args, t_fix=None):
    """eval_cond_predictor.py:34-89: mean absolute error (rescaled by the dataset std) over a loader of
    (x, node_mask, edge_mask, node_features, y) batches."""
    losses, errors = [], []
    for x, node_mask, edge_mask, node_features, y in dataloader:
        x = _to_numpy(x).astype(np.float32)
        nm = _to_numpy(node_mask).astype(np.float32)
        nm3 = nm.reshape(x.shape[0], x.shape[1], 1)
        # remove_mean_with_mask (utils.py:33-44)
        x = x - (x * nm3).sum(1, keepdims=True) / np.maximum(nm3.sum(1, keepdims=True), 1) * nm3
        loss, err = compute_loss(cond_predictor, x, node_features, nm, edge_mask, y, edm_model, edm_args, t_fix)
        losses.append(float(loss))
        errors.append(_to_numpy(err))
    std = _to_numpy(dataloader.dataset.std).astype(np.float32)
    print(f"[{tag}] loss: {np.mean(losses):.4f}+-{np.std(losses):.4f}")
    return float((np.concatenate(errors) * std[None, :]).mean())


def t_sweep(cond_predictor, edm_model, dataloader, args, edm_args, n_points=11):
    """eval_cond_predictor.main (eval_cond_predictor.py:92-113) without the plot -> (times, MAE per time)."""
    times = np.linspace(0, edm_model.T, n_points)
    return times, [val_epoch("test", cond_predictor, edm_model, dataloader, args, edm_args, t_fix=t) for t in times]
