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
    """Factorised algorithm, live edges / live nodes only, unpadded H: the work the kernel NEEDS to do."""
    E, Nn = n_live_edges, n_nodes
    H, L, S = edm["nf"], edm["n_layers"], edm.get("inv_sublayers", 1)
    gcl = 2 * Nn * 2 * H * H + 2 * E * (H * H + 5 * H) + 2 * Nn * 3 * H * H
    equ = 2 * Nn * 2 * H * H + 2 * E * (H * H + 5 * H)
    f = L * (S * gcl + equ) + 4 * Nn * (F + 1) * H
    if pred is not None:
        H, L = pred["nf"], pred["n_layers"]
        fwd = 2 * Nn * 2 * H * H + 2 * E * (2 * H * H + 6 * H) + 2 * Nn * 3 * H * H
        # reverse pass (pre-activations come back from the forward's stash, nothing is recomputed): per edge the two
        # transposed GEMMs Wc1^T, W2^T; per node Wn2^T, Wn1h^T, Wn1a^T, A^T, B^T.  The last layer has no coordinate branch.
        bwd = 2 * E * (2 * H * H + 6 * H) + 2 * Nn * 5 * H * H
        last = 2 * E * (H * H + H)  # Wc1 (forward) and Wc1^T (reverse) of the last layer are never needed
        f += L * (fwd + bwd) - 2 * last + 4 * Nn * ((F + 1) * H + H * K)
    return f


def pad16(h):
    return (h + 15) // 16 * 16


def _pad_hidden_kernel(h):
    for s in (32, 48, 64, 128, 192, 208, 256):
        if s >= h:
            return s
    raise ValueError(h)


def step_mfma_issued(edge_units, ncols, edm, pred=None, variant="w4"):
    """v_mfma_f32_16x16x4_f32 instructions (2048 FLOP each) ONE molecule issues per reverse step, counted from the kernel's
    loop structure (validated against SQ_INSTS_VALU_MFMA_MOPS_F32 / 4: profiles/*pmc_summary.csv).

    variant "w4" (4 waves, 32-edge passes):  edge_units = list of 32-edge passes per wave [4]
    variant "w8" (8 waves, 16-edge tiles):   edge_units = number of 16-edge tiles of the molecule
    ncols = node columns the node-level GEMMs produce (16-column tiles, pairs of tiles beyond 16)."""
    ncols = int(ncols)
    nt = 1 if ncols <= 16 else 2 * (((ncols + 15) // 16 + 1) // 2)
    if variant == "w4":
        waves, pairs = 4, int(sum(int(v) for v in edge_units))
        tiles16 = 2 * pairs
    else:
        waves, tiles16 = 8, int(edge_units)

    def node(T, kt):  # K chunks x 4 k-steps x output tiles (the 4-wave kernels recompute a tile in idle tile slots)
        return kt * 4 * ((-(-T // waves)) * waves if variant == "w4" else T) * nt

    def edge(T, nf=0):  # one 16-edge tile through one T x T matrix; the 8-wave kernels issue a K tail (nf % 16 == 4) as 1 k-step
        ksteps = 4 * T - 3 if (variant != "w4" and nf % 16 == 4 and 16 * T - nf == 12) else 4 * T
        return T * ksteps

    Te = _pad_hidden_kernel(edm["nf"]) // 16
    L, S = edm["n_layers"], edm.get("inv_sublayers", 1)
    n = L * (S * (node(Te, 5 * Te) + tiles16 * edge(Te, edm["nf"])) + node(Te, 2 * Te) + tiles16 * edge(Te, edm["nf"]))
    if pred is not None:
        Tp = _pad_hidden_kernel(pred["nf"]) // 16
        Lp = pred["n_layers"]
        fwd = Lp * (node(Tp, 5 * Tp) + tiles16 * edge(Tp, pred["nf"])) + (Lp - 1) * tiles16 * edge(Tp, pred["nf"])
        n += 2 * fwd  # the reverse pass issues the same counts with the transposed matrices
    return n


def step_bytes_fused(B, N, F, weight_bytes, stash_bytes_per_mol=0):
    """HBM bytes a fully fused per-timestep launch must touch: state in/out + weights once
    (+ the predictor's per-layer node stash, written and read once)."""
    return 2 * B * N * (3 + F) * 4 + weight_bytes + 2 * B * stash_bytes_per_mol
