"""FLOP / byte accounting for the roofline numbers (2 FLOP per MAC).

``as_written``: what the reference executes -- dense N x N edge set incl. self loops and masked
pairs, concat + Linear(2H+2 -> H) per edge.  This is SURVEY 8(d)'s per-unit figure (C3: 1.464 GFLOP
per molecule-step) and the numerator of ``roofline.achieved``.
``useful``: the factorised algorithm the kernel runs (W1 [h_i|h_j|r|d0] = A h_i + B h_j + ...,
computed per node), live edges only, unpadded H.
``issued``: what the matrix cores are actually asked to do (features padded to 16, nodes to 16,
each wave's edge list to 32, backward one 16-edge tile at a time).
"""
from __future__ import annotations


def _edm_as_written(N, H, F, L, S):
    E, Nn = N * N, N
    gcl = 2 * E * ((2 * H + 2) * H + H * H + H) + 2 * Nn * 3 * H * H
    equ = 2 * E * ((2 * H + 2) * H + H * H + H)
    return L * (S * gcl + equ) + 4 * Nn * (F + 1) * H


def _pred_as_written(N, H, F, L, K):
    E, Nn = N * N, N
    lay = 2 * E * ((2 * H + 2) * H + H * H + H) + 2 * E * (H * H + H) + 2 * Nn * 3 * H * H
    return L * lay + 2 * Nn * ((F + 1) * H + H * K)


def step_flops_as_written(N, F, edm, pred=None, K=5):
    """FLOPs of one reverse step for ONE molecule as the reference writes it; guided = EDM + 2 x predictor."""
    f = _edm_as_written(N, edm["nf"], F, edm["n_layers"], edm.get("inv_sublayers", 1))
    if pred is not None:
        f += 2 * _pred_as_written(N, pred["nf"], F, pred["n_layers"], K)
    return f


def step_flops_useful(n_live_edges, n_nodes, F, edm, pred=None, K=5):
    """Factorised algorithm, live edges / live nodes only."""
    E, Nn = n_live_edges, n_nodes
    H, L, S = edm["nf"], edm["n_layers"], edm.get("inv_sublayers", 1)
    gcl = 2 * Nn * 2 * H * H + 2 * E * (H * H + 5 * H) + 2 * Nn * 3 * H * H
    equ = 2 * Nn * 2 * H * H + 2 * E * (H * H + 5 * H)
    f = L * (S * gcl + equ) + 4 * Nn * (F + 1) * H
    if pred is not None:
        H, L = pred["nf"], pred["n_layers"]
        fwd = 2 * Nn * 2 * H * H + 2 * E * (2 * H * H + 6 * H) + 2 * Nn * 3 * H * H
        # reverse pass: recompute (W2, Wc1) + transposed (Wc1^T, W2^T) per edge; node level: recompute 5, backward 5
        bwd = 2 * E * (4 * H * H + 10 * H) + 2 * Nn * 10 * H * H
        f += L * (fwd + bwd) + 4 * Nn * ((F + 1) * H + H * K)
    return f


def step_bytes_fused(B, N, F, weight_bytes, stash_bytes_per_mol=0):
    """HBM bytes a fully fused per-timestep launch must touch: state in/out + weights once
    (+ the predictor's per-layer node stash, written and read once)."""
    return 2 * B * N * (3 + F) * 4 + weight_bytes + 2 * B * stash_bytes_per_mol
