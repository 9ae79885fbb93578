"""Host mirror of sampling_edm.py:119-224: mask construction + the two sampling entry points with the
reference's signatures and return values (x, one_hot, node_mask, edge_mask)."""
from __future__ import annotations

import numpy as np

from ._lib import GaudiError
from .models_edm import _like_ref, _to_numpy


def node2edge_mask(node_mask):
    """sampling_edm.py:119-125: [B,N] -> [B,N,N], zero diagonal."""
    nm = _to_numpy(node_mask).astype(np.float32)
    em = nm[:, None, :] * nm[:, :, None]
    em = em * (1.0 - np.eye(nm.shape[1], dtype=np.float32))[None]
    return em


def build_masks(nodesxsample, max_nodes: int, orientation: bool):
    """node_mask [B,N,1], edge_mask [B*N*N,1] exactly as sampling_edm.py:135-161 / 176-209 build them:
    with ``orientation`` (dataset != 'cata') every ring gets an orientation node, N doubles, and the
    ring<->orientation identity blocks stay UNMASKED for padded rings (reference behaviour)."""
    n = np.asarray(_to_numpy(nodesxsample)).astype(np.int64)
    B = len(n)
    nm = np.zeros((B, max_nodes), np.float32)
    for i in range(B):
        nm[i, : n[i]] = 1
    em = node2edge_mask(nm)
    N = max_nodes
    if orientation:
        eye = np.broadcast_to(np.eye(N, dtype=np.float32), (B, N, N))
        top = np.concatenate([em, eye], axis=1)
        right = np.broadcast_to(np.concatenate([np.eye(N, dtype=np.float32), np.zeros((N, N), np.float32)], 0),
                                (B, 2 * N, N))
        em = np.concatenate([top, right], axis=2)
        nm = np.concatenate([nm, nm], axis=1)
        N *= 2
    return nm[:, :, None].copy(), np.ascontiguousarray(em).reshape(-1, 1), N


def _check(x, node_mask):
    """assert_correctly_masked + assert_mean_zero_with_mask (edm/equivariant_diffusion/utils.py:52-65)."""
    x = _to_numpy(x)
    nm = _to_numpy(node_mask)
    assert np.abs(x * (1 - nm)).max() < 1e-4, "Variables not masked properly."
    largest = np.abs(x).max()
    err = np.abs(x.sum(axis=1, keepdims=True)).max()
    rel = err / (largest + 1e-10)
    assert rel < 1e-2, f"Mean is not zero, relative_error {rel}"


def sample_pos_edm(args, model, nodesxsample, std=0.7):
    """sampling_edm.py:128-169: unconditional molecules, padded to args.max_nodes."""
    n = np.asarray(_to_numpy(nodesxsample)).astype(np.int64)
    max_nodes = int(args.max_nodes)
    assert int(n.max()) <= max_nodes
    nm, em, N = build_masks(n, max_nodes, args.dataset != "cata")
    x, h = model.sample(len(n), N, nm, em, std=std)
    _check(x, nm)
    return x, h["categorical"], _like_ref(nm), _like_ref(em)


def sample_guidance(args, model, target_function, nodesxsample, scale=1, std=1.0):
    """sampling_edm.py:172-224: guided molecules, padded to the batch maximum."""
    n = np.asarray(_to_numpy(nodesxsample)).astype(np.int64)
    nm, em, N = build_masks(n, int(n.max()), args.dataset != "cata")
    x, h = model.sample_guidance(len(n), target_function, nm, em, scale, fix_noise=False, std=std)
    _check(x, nm)
    return x, h["categorical"], _like_ref(nm), _like_ref(em)
