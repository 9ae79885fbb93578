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


FLOP_MFMA_F32 = 2048      # v_mfma_f32_16x16x4_f32: 16 x 16 x 4 MACs
FLOP_MFMA_BF16 = 16384    # v_mfma_f32_16x16x32_bf16: 16 x 16 x 32 MACs
PEAK_F32_TFLOPS = 157.3   # 256 CUs x 4 SIMDs x 64 FLOP/clk x 2.4 GHz (MI355X_MICROARCH.md)
PEAK_BF16_TFLOPS = 2516.6  # ... x 1024 FLOP/clk (dense)


def _has_ktail(nf, HP):
    return nf % 16 == 4 and HP - nf == 12


def step_mfma_counts(edge_units, ncols, edm, pred=None, variant="w4", max_col_tiles=2):
    """(fp32, bf16) matrix instructions ONE molecule issues per reverse step, counted from the kernels' loop structure
    (validated against SQ_INSTS_VALU_MFMA_MOPS_F32 / 4 and SQ_INSTS_VALU_MFMA_MOPS_BF16 / 32: profiles/*pmc_summary.csv).
    fp32 = v_mfma_f32_16x16x4_f32 EQUIVALENTS (1 024 MACs, 32 matrix-pipe cycles on a SIMD): a 4x4x1_16B instruction of the
    node GEMMs' tail tile counts 0.25.

    variant "w4"  (4 waves, 32-edge passes):  edge_units = list of 32-edge passes per wave [4]
    variant "w8"  (8 waves, 16-edge tiles, fp32 MFMAs):        edge_units = number of 16-edge tiles of the molecule
    variant "w8s" (8 waves, edge GEMMs on fp16-pair operands: three v_mfma_f32_16x16x32_f16 per output tile and 32 inputs,
                   a K tail as one fp32 k-step per tile; node GEMMs on fp16 pairs: three v_mfma_f32_16x16x32_f16 -- the same
                   rate, counted with the bf16 instructions -- per output-tile slot, column tile and 32 inputs)
    ncols = node columns the node-level GEMMs produce (16-column tiles, pairs of tiles beyond 16; max_col_tiles = 3 -- the V8G
    kernels -- runs 33..48 columns as one pass over three tiles, w8_common.h: node_gemm_n)."""
    ncols = int(ncols)
    nt = 1 if ncols <= 16 else (3 if max_col_tiles >= 3 and 32 < ncols <= 48 else 2 * (((ncols + 15) // 16 + 1) // 2))
    if variant == "w4":
        waves, pairs = 4, int(sum(int(v) for v in edge_units))
        tiles16 = 2 * pairs
    else:
        waves, tiles16 = 8, int(edge_units)

    def node(T, kt, nf):  # -> (fp32, 16-bit) instructions of kt / T node-level matrices
        if variant == "w4":  # K chunks x 4 k-steps x output tiles (the 4-wave kernels recompute a tile in idle tile slots)
            return kt * 4 * (-(-T // waves)) * waves * nt, 0
        if variant == "w8s":
            # fp16-pair node GEMMs (w8_nodes_f16.h, round 5): three v_mfma_f32_16x16x32_f16 per chunk of 32 inputs, column tile and
            # output-tile SLOT -- every wave runs the slots of the widest wave (two for more than 8 tiles; a slot without a tile
            # multiplies zeros) -- and, for an odd tile count, the last 16 inputs as fp32 steps (one when H % 16 == 4, else four)
            slots = waves * (2 if T > waves else 1)
            mats = kt // T
            tail = (1 if _has_ktail(nf, 16 * T) else 4) if T % 2 else 0
            return mats * slots * tail * nt, mats * (T // 2) * 3 * slots * nt
        # 8-wave fp32 kernels, widths with a tail tile (196 -> 208, 36 -> 48): that tile issues v_mfma_f32_4x4x1_16B_f32, a quarter
        # of a 16x16x4 instruction each in MACs and in matrix-pipe cycles (w8_common.h: tail_lane) -> 0.25 instruction-equivalents
        tail44 = _has_ktail(nf, 16 * T) and 16 * T in (208, 48)
        return kt * 4 * (T - 0.75 if tail44 else T) * nt, 0

    def edge(T, nf):  # one 16-edge tile through one T x T matrix -> (fp32, bf16)
        tail = variant != "w4" and _has_ktail(nf, 16 * T)
        if variant == "w8s":
            tail = tail and T % 2 == 1 and T >= 3
            nc = (T + 1) // 2 - (1 if tail else 0)
            return (T if tail else 0), nc * T * 3  # (round 5: fp16 pairs, three piece products; rounds 2-4: bf16 x 3, six)
        return T * (4 * T - 3 if tail else 4 * T), 0

    f32 = bf = 0
    Te = _pad_hidden_kernel(edm["nf"]) // 16
    L, S = edm["n_layers"], edm.get("inv_sublayers", 1)
    e32, ebf = edge(Te, edm["nf"])
    n32, n16 = node(Te, (5 * S + 2) * Te, edm["nf"])  # 5 node-level matrices per GCL, 2 per EquivariantUpdate
    f32 += L * (n32 + (S + 1) * tiles16 * e32)
    bf += L * (n16 + (S + 1) * tiles16 * ebf)
    if pred is not None:
        Tp = _pad_hidden_kernel(pred["nf"]) // 16
        Lp = pred["n_layers"]
        e32, ebf = edge(Tp, pred["nf"])
        n_edge = 2 * (2 * Lp - 1)  # W2 + Wc1 per layer (no Wc1 in the last), the same again transposed in the reverse pass
        n32, n16 = node(Tp, 5 * Tp, pred["nf"])
        f32 += 2 * Lp * n32 + n_edge * tiles16 * e32
        bf += 2 * Lp * n16 + n_edge * tiles16 * ebf
    return f32, bf


def step_mfma_issued(edge_units, ncols, edm, pred=None, variant="w4", max_col_tiles=2):
    """fp32 matrix instructions per molecule-step of the fp32-MFMA kernel families (see step_mfma_counts)."""
    return step_mfma_counts(edge_units, ncols, edm, pred, variant, max_col_tiles)[0]


def step_weight_stream_bytes(edm, pred=None, variant="w8s", rounds=1):
    """Bytes of packed weights ONE workgroup streams from L2 per reverse step: every node-level matrix as (16 T)^2 fp32
    (lane-linear tiles), every edge-level matrix as fp32 tiles or, for "w8s", as its split image (units of 1 KiB: K chunks x
    output tiles x 2 fp16 pieces; a K tail as T fp32 tiles; the node-level images -- fp16 pairs too -- have the fp32 size).  Same matrix counts as step_mfma_counts.  No reuse across
    workgroups inside a CU -- one molecule (or packed group) per workgroup -- so a launch of G workgroups moves G times this
    through the L2 -> CU fabric per step.  rounds: rounds of eight edge tiles of the workgroup's graph (wide groups: 2) -- an
    edge-level matrix is streamed once per round, a node-level one once per workgroup."""
    def node_b(T):
        return (16 * T) ** 2 * 4

    def edge_b(T, nf):
        if variant != "w8s":
            return (16 * T) ** 2 * 4
        tail = _has_ktail(nf, 16 * T) and T % 2 == 1 and T >= 3
        nc = (T + 1) // 2 - (1 if tail else 0)
        return nc * T * 2 * 1024 + (T * 1024 if tail else 0)

    Te = _pad_hidden_kernel(edm["nf"]) // 16
    L, S = edm["n_layers"], edm.get("inv_sublayers", 1)
    total = L * ((S * 5 + 2) * node_b(Te) + rounds * (S + 1) * edge_b(Te, edm["nf"]))
    if pred is not None:
        Tp = _pad_hidden_kernel(pred["nf"]) // 16
        Lp = pred["n_layers"]
        total += 2 * Lp * 5 * node_b(Tp) + rounds * 2 * (2 * Lp - 1) * edge_b(Tp, pred["nf"])
    return total


def step_bytes_fused(B, N, F, weight_bytes, stash_bytes_per_mol=0):
    """HBM bytes a fully fused per-timestep launch must touch: state in/out + weights once
    (+ the predictor's per-layer node stash, written and read once)."""
    return 2 * B * N * (3 + F) * 4 + weight_bytes + 2 * B * stash_bytes_per_mol
