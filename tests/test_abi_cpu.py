"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol declared in
include/gaudi_hip.h; no compute calls (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from gaudi_amd import _lib, build
    build.build()
    lib = _lib.load_library()
    header = open(os.path.join(ROOT, "include", "gaudi_hip.h")).read()
    declared = set(re.findall(r"\b(gaudi_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in gaudi_hip.h but not exported"
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gaudi_amd._lib import GaudiError
    from gaudi_amd.engine import Engine
    with pytest.raises(GaudiError):
        Engine(0)


# ---- device-free host logic of the library (C++), checked on CPU -------------------------------------------------
import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402


def _lib():
    from gaudi_amd import _lib, build
    build.build()
    return _lib.load_library(), _lib


def test_host_schedule_matches_reference(golden):
    """gamma table and per-step scalars built by the C++ host code vs the reference's PredefinedNoiseSchedule /
    sigma_and_alpha_t_given_s outputs (golden g1)."""
    lib, L = _lib()
    g = golden("g1_schedule")
    for T in (50, 1000):
        gamma = np.empty(T + 1, np.float32)
        coef = np.empty((T, 4), np.float32)
        assert lib.gaudi_host_schedule(T, 2.0, 1e-5, L.fptr(gamma), L.fptr(coef)) == 0
        np.testing.assert_allclose(gamma, g[f"gamma_T{T}"], rtol=2e-7, atol=0)
        for row in g[f"coef_T{T}"]:
            s = int(row[0])
            np.testing.assert_allclose(coef[s], [row[1], row[3], row[4], row[7]], rtol=1e-5)
    assert lib.gaudi_host_schedule(0, 2.0, 1e-5, L.fptr(gamma), None) != 0
    # the 'cosine' schedule (noise_power = 0; en_diffusion.py:64-81) against the reference's tables (golden g20)
    g = golden("g20_cosine_and_mean")
    for T in (50, 1000):
        gamma = np.empty(T + 1, np.float32)
        coef = np.empty((T, 4), np.float32)
        assert lib.gaudi_host_schedule(T, 0.0, 1e-5, L.fptr(gamma), L.fptr(coef)) == 0
        np.testing.assert_allclose(gamma, g[f"gamma_T{T}"], rtol=3e-7, atol=0)
        for row in g[f"coef_T{T}"]:
            s = int(row[0])
            np.testing.assert_allclose(coef[s], [row[1], row[3], row[4], row[7]], rtol=2e-5)
    assert lib.gaudi_host_schedule(50, -1.0, 1e-5, L.fptr(gamma), None) != 0


def _meta(lib, L, nm, em):
    B, N = nm.shape[0], nm.shape[1]
    nm = np.ascontiguousarray(nm.reshape(B, N), np.float32)
    em = np.ascontiguousarray(em.reshape(B, N, N), np.float32)
    ew = C.c_int32()
    order = np.empty(B, np.int32)
    npairs = np.empty((B, 4), np.int32)
    seg = np.empty((B, N), np.uint32)
    cap = B * 4 * 4096
    edges = np.empty(cap, np.uint32)
    emask = np.empty(cap, np.float32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    up = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint32))
    ncols = np.empty(B, np.int32)
    rc = lib.gaudi_host_graph_meta(B, N, L.fptr(nm), L.fptr(em), C.byref(ew), ip(order), ip(npairs), up(seg), up(edges),
                                   L.fptr(emask), cap, ip(ncols))
    assert rc == 0
    EW = ew.value
    return EW, order, npairs, seg, edges[: B * 4 * EW].reshape(B, 4, EW), emask[: B * 4 * EW].reshape(B, 4, EW), ncols


@pytest.mark.parametrize("kind", ["cata", "hetro", "random"])
def test_host_graph_meta_invariants(kind):
    """Every live edge appears exactly once, in the list of the wave that owns its receiving node, lists are sorted by
    receiving node, padded to 32 with mask-0 slots, and the launch order is heaviest-first.  Live = edge_mask != 0 and
    not both endpoints masked (the reference's unmasked identity edges between a padded ring and its orientation node
    are dropped: they reach no live node); ncols = 1 + last node that is live or touches a live edge."""
    from oracle import gaudi_oracle as O
    lib, L = _lib()
    rng = np.random.default_rng(3)
    if kind == "cata":
        nm, em = O.build_masks([4, 11, 7, 1, 11, 0], 11, False)
    elif kind == "hetro":
        nm, em = O.build_masks([3, 5, 10, 7], 10, True)
    else:
        B, N = 5, 13
        nm = np.ones((B, N, 1), np.float32)
        em = (rng.random((B, N, N)) < 0.3).astype(np.float32) * rng.choice([1.0, 0.5], size=(B, N, N))
        em[2] = 0
    B, N = nm.shape[0], nm.shape[1]
    em3 = np.asarray(em, np.float32).reshape(B, N, N)
    EW, order, npairs, seg, edges, emask, ncols = _meta(lib, L, nm, em3)
    assert EW % 32 == 0 and EW >= 32
    nmf = nm.reshape(B, N) != 0
    live3 = (em3 != 0) & (nmf[:, :, None] | nmf[:, None, :])
    if kind == "hetro":
        assert (live3 != (em3 != 0)).any()  # the fixture does contain padded-ring <-> orientation identity edges
    totals = live3.reshape(B, -1).sum(1)
    assert sorted(order.tolist()) == list(range(B))
    assert all(totals[order[k]] >= totals[order[k + 1]] for k in range(B - 1))
    for b in range(B):
        seen = {}
        for w in range(4):
            n_slots = npairs[b, w] * 32
            assert n_slots <= EW
            ii = (edges[b, w, :n_slots] & 255).astype(int)
            jj = ((edges[b, w, :n_slots] >> 8) & 255).astype(int)
            mk = emask[b, w, :n_slots]
            live = mk != 0
            assert np.all(np.diff(ii[live]) >= 0)  # sorted by receiving node
            assert n_slots - live.sum() < 32 or live.sum() == 0  # padding only up to the next multiple of 32
            for i, j, m in zip(ii[live], jj[live], mk[live]):
                assert (i, j) not in seen
                seen[(i, j)] = m
                assert seg[b, i] >> 30 == w  # owned by this wave
        want = {(i, j): em3[b, i, j] for i in range(N) for j in range(N) if live3[b, i, j]}
        assert seen == want
        touched = nmf[b] | live3[b].any(0) | live3[b].any(1)
        assert ncols[b] == (int(np.nonzero(touched)[0].max()) + 1 if touched.any() else 1)
        for n in range(N):  # segment word: start/len of node n's run inside its wave's list
            w, st, ln = seg[b, n] >> 30, (seg[b, n] >> 15) & 0x7FFF, seg[b, n] & 0x7FFF
            assert ln == int(live3[b, n].sum())
            if ln:
                assert np.all((edges[b, w, st:st + ln] & 255) == n)


def test_host_pack_matrix_layout():
    lib, L = _lib()
    H, HP = 36, 48
    W = np.random.default_rng(0).standard_normal((H, 2 * H + 2)).astype(np.float32)
    for tr in (0, 1):
        out = np.empty(HP * HP, np.float32)
        assert lib.gaudi_host_pack_matrix(H, 2 * H + 2, H, HP, tr, L.fptr(W), L.fptr(out)) == 0
        blk = W[:, H:2 * H].T if tr else W[:, H:2 * H]  # logical [o][k]
        T = HP // 16
        p = out.reshape(T, T, 16, 16)  # [k/16][o/16][o%16][k%16]
        full = np.zeros((HP, HP), np.float32)
        full[:H, :H] = blk
        got = p.transpose(1, 2, 0, 3).reshape(HP, HP)
        assert np.array_equal(got, full)


def _f16_to_f32(u16):
    return u16.view(np.float16).astype(np.float32)


def test_host_pack_matrix_split_is_laid_out_as_documented():
    """The fp16-pair image the default edge GEMMs stream (csrc/w8_split.h): two fp16 pieces of w * scale that reproduce it to 22
    significant bits (to fp16's absolute floor for entries far below the largest), in the documented unit / lane / slot
    positions; a K tail (H % 16 == 4, odd tile count) as fp32 tiles of w * scale."""
    lib, L = _lib()
    rng = np.random.default_rng(1)
    for H, HP, ktail in ((36, 48, 1), (36, 48, 0), (60, 64, 0), (20, 32, 1), (196, 208, 1)):
        T, NC = HP // 16, (HP // 16 + 1) // 2
        W = (rng.standard_normal((H, H + 3)) * rng.choice([1e-3, 1.0, 30.0], size=(H, H + 3))).astype(np.float32)
        blocks = (L.FP * 1)(L.fptr(W))
        scale = np.zeros(1, np.float32)
        one = lambda v: np.array([v], np.int32).ctypes.data_as(L.IP)
        assert lib.gaudi_host_weight_scale(1, blocks, one(H), one(H + 3), one(H + 3), L.fptr(scale)) == 0
        sc = float(scale[0])
        assert 2.0 ** 13 <= np.abs(W).max() * sc < 2.0 ** 14 and np.log2(sc) == round(np.log2(sc))
        for tr in (0, 1):
            out = np.empty(NC * T * 2 * 256, np.float32)
            assert lib.gaudi_host_pack_matrix_split(H, H + 3, 2, HP, tr, ktail, sc, L.fptr(W), L.fptr(out)) == 0
            blk = W[:, 2:2 + H].T if tr else W[:, 2:2 + H]  # logical [o][k]
            full = np.zeros((HP, HP), np.float32)
            full[:H, :H] = blk * np.float32(sc)
            tail = bool(ktail) and HP - H == 12 and T % 2 == 1 and T >= 3
            units16 = out.view(np.uint16).reshape(NC, T, 2, 64, 8)
            units32 = out.reshape(NC, T, 2, 64, 4)
            for m in range(NC):
                for t in range(T):
                    if tail and m == NC - 1:
                        tile = units32.reshape(NC, T * 2, 64, 4)[m, t]  # fp32 tile t of the tail chunk: [L][4], element 0
                        for kk in range(4):
                            assert np.array_equal(tile[kk * 16:(kk + 1) * 16, 0], full[16 * t:16 * t + 16, 16 * (T - 1) + kk])
                        continue
                    hi, lo = _f16_to_f32(units16[m, t, 0]), _f16_to_f32(units16[m, t, 1])  # [64 lanes][8 slots]
                    for Ln in range(64):
                        row, g = Ln & 15, Ln >> 4
                        for e in range(8):
                            k = 16 * (2 * m + (e >> 2)) + 4 * g + (e & 3)
                            want = full[16 * t + row, k] if k < HP else np.float32(0)
                            assert hi[Ln, e] == np.float32(np.float16(want)), (H, tr, m, t, Ln, e)
                            assert lo[Ln, e] == np.float32(np.float16(want - hi[Ln, e]))
                            err = abs(float(hi[Ln, e]) + float(lo[Ln, e]) - float(want))
                            assert err <= max(abs(float(want)) * 2.0 ** -22, 2.0 ** -25)


def test_host_pack_matrix_f16_node_image():
    """The fp16-pair image of a node-GEMM matrix (csrc/w8_nodes_f16.h): units [K chunk of 32][output tile][piece], lane (row, 8
    inputs); piece 1 carries the remainder times 2^11; an odd tile count leaves the last 16 inputs as unscaled fp32 k-steps."""
    lib, L = _lib()
    rng = np.random.default_rng(4)
    for H, HP in ((36, 48), (48, 48), (60, 64), (196, 208), (192, 192)):
        T, nc = HP // 16, HP // 32
        W = (rng.standard_normal((H, H + 5)) * rng.choice([1e-4, 1.0, 5.0], size=(H, H + 5))).astype(np.float32)
        sc = 2.0 ** 11
        for tr in (0, 1):
            out = np.empty(HP * HP, np.float32)
            assert lib.gaudi_host_pack_matrix_f16(H, H + 5, 3, HP, tr, sc, L.fptr(W), L.fptr(out)) == 0
            blk = W[:, 3:3 + H].T if tr else W[:, 3:3 + H]
            full = np.zeros((HP, HP), np.float32)
            full[:H, :H] = blk
            units16 = out[:nc * T * 512].view(np.uint16).reshape(nc, T, 2, 64, 8)
            for m in range(nc):
                for t in range(T):
                    hi, lo = _f16_to_f32(units16[m, t, 0]), _f16_to_f32(units16[m, t, 1])
                    for Ln in range(64):
                        row, g = Ln & 15, Ln >> 4
                        want = full[16 * t + row, 32 * m + 8 * g:32 * m + 8 * g + 8] * np.float32(sc)
                        assert np.array_equal(hi[Ln], want.astype(np.float16).astype(np.float32))
                        rec = hi[Ln].astype(np.float64) + lo[Ln].astype(np.float64) / 2048
                        assert np.all(np.abs(rec - want) <= np.maximum(np.abs(want) * 2.0 ** -22, 2.0 ** -36))
            if T % 2:  # the last half chunk: [tile][k-step q][lane (row, g)] = w[row][16 (T-1) + 4 q + g], unscaled
                tail = out[nc * T * 512:nc * T * 512 + T * 256].reshape(T, 4, 4, 16)
                for t in range(T):
                    for q in range(4):
                        for g in range(4):
                            assert np.array_equal(tail[t, q, g], full[16 * t:16 * t + 16, 16 * (T - 1) + 4 * q + g])
            else:
                assert nc * T * 512 == HP * HP


def test_host_weight_scale_refusals():
    """NodeScale (gaudi_hip.hip): an infinite weight, or a matrix whose largest entry lies more than 2^17 below the largest of
    all (edge-level matrices; round 6 -- 2^12 through round 5), refuses the fp16 images (scale 0: the network runs the
    fp32-instruction kernels); NaN entries are ignored."""
    lib, L = _lib()
    rng = np.random.default_rng(5)
    a = rng.standard_normal((8, 8)).astype(np.float32)
    b = (rng.standard_normal((8, 8)) * 1e-2).astype(np.float32)

    def scale(*mats):
        blocks = (L.FP * len(mats))(*[L.fptr(m) for m in mats])
        arr = lambda v: np.array(v, np.int32).ctypes.data_as(L.IP)
        out = np.zeros(1, np.float32)
        assert lib.gaudi_host_weight_scale(len(mats), blocks, arr([m.shape[0] for m in mats]), arr([m.shape[1] for m in mats]),
                                           arr([m.shape[1] for m in mats]), L.fptr(out)) == 0
        return float(out[0])

    assert scale(a, b) > 0
    c = a.copy()
    c[2, 3] = np.nan
    assert scale(c, b) == scale(a, b)
    c[1, 1] = np.inf
    assert scale(c, b) == 0
    assert scale(a, (b * 1e-3).astype(np.float32)) == scale(a, b)  # 1e-5 of the largest matrix: refused through round 5
    assert scale(a, (b * 1e-5).astype(np.float32)) == 0  # 1e-7: beyond 2^-17


def test_node_operand_layout_is_bank_conflict_free():
    """csrc/w8_nodes_f16.h: nh_bpos -- the LDS position of (node column c, input group g) inside a B unit of the fp16-pair node GEMMs.
    Readers: the matrix instructions' B operands, 16 bytes per lane, lanes (c = 0..15) of one g in one pass -> must be 256
    consecutive bytes.  Writers: a row's split, 8 bytes per lane, one pass = 4 chunks x 4 g x 2 halves of one column -> 64
    different banks (4-byte banks, 64 of them).  Every (chunk, tile, c, g) owns its own 16 bytes.  (the
    column-major unit of the round's first version put four reading lanes on every bank.)"""
    lib, L = _lib()

    def off(nct, chunk, tile, c, g):
        out = np.zeros(1, np.int32)
        assert lib.gaudi_host_node_operand_offset(nct, chunk, tile, c, g, out.ctypes.data_as(L.IP)) == 0
        return int(out[0])

    for nct in (1, 2, 3):
        seen = set()
        for chunk in range(7):
            for tile in range(nct):
                for g in range(4):
                    lanes = sorted(off(nct, chunk, tile, c, g) for c in range(16))
                    assert lanes == list(range(lanes[0], lanes[0] + 64, 4)) and lanes[0] % 4 == 0, (nct, chunk, tile, g, lanes)
                    for c in range(16):
                        o = off(nct, chunk, tile, c, g)
                        assert o % 4 == 0 and o not in seen
                        seen.add(o)
        # piece 0 and piece 1 (256 floats further) of a tile never overlap another tile or chunk
        spans = sorted((o, o + 4) for o in seen) + sorted((o + 256, o + 260) for o in seen)
        spans.sort()
        assert all(a[1] <= b[0] for a, b in zip(spans, spans[1:]))
    # store pass of one row (nct = 1, where the chunk stride is padded for it): chunks m..m+3, all g, both halves -> 64 banks
    for c in range(16):
        for m0 in (0, 3):
            banks = []
            for m in range(m0, m0 + 4):
                for g in range(4):
                    for half in range(2):
                        o = off(1, m, 0, c, g) + 2 * half
                        banks += [o % 64, (o + 1) % 64]
            assert len(set(banks)) == 64, (c, m0)
    out = np.zeros(1, np.int32)
    assert lib.gaudi_host_node_operand_offset(4, 0, 0, 0, 0, out.ctypes.data_as(L.IP)) != 0
    assert lib.gaudi_host_node_operand_offset(1, 0, 0, 16, 0, out.ctypes.data_as(L.IP)) != 0


def test_host_split_keeps_non_finite_weights_non_finite():
    """A NaN (any payload, either sign) planted in an edge-GEMM matrix must stay a NaN in every fp16 piece of the split image; an
    infinite weight gives inf in the high piece and NaN = inf - inf below it (the loaders refuse such a network: NodeScale)."""
    lib, L = _lib()
    H, HP = 36, 48
    T, NC = HP // 16, (HP // 16 + 1) // 2
    W = np.random.default_rng(2).standard_normal((H, H)).astype(np.float32)
    payloads = [0xFFFFFFFF, 0x7F800001, 0x7FC00000, 0xFFC00001, 0x7FFFFFFF, 0xFF800001]
    spots = [(0, 0), (3, 17), (20, 5), (35, 31), (7, 7), (16, 16)]
    Wv = W.view(np.uint32)
    for (o, k), bits in zip(spots, payloads):
        Wv[o, k] = bits
    W[1, 2], W[9, 30] = np.inf, -np.inf
    out = np.empty(NC * T * 2 * 256, np.float32)
    assert lib.gaudi_host_pack_matrix_split(H, H, 0, HP, 0, 0, 1024.0, L.fptr(W), L.fptr(out)) == 0
    units16 = out.view(np.uint16).reshape(NC, T, 2, 64, 8)

    def pieces(o, k):
        tile, g, e = k // 16, (k % 16) // 4, 4 * ((k // 16) & 1) + (k & 3)
        return _f16_to_f32(units16[tile // 2, o // 16, :, g * 16 + o % 16, e])

    for o, k in spots:
        assert np.isnan(pieces(o, k)).all(), (o, k, pieces(o, k))
    for (o, k), sgn in (((1, 2), 1.0), ((9, 30), -1.0)):
        p = pieces(o, k)
        assert np.isinf(p[0]) and np.sign(p[0]) == sgn and not np.isfinite(p).any()
    fin = np.isfinite(W)
    for o in range(H):
        for k in range(H):
            if fin[o, k]:
                p = pieces(o, k)
                assert abs(float(p[0]) + float(p[1]) - float(W[o, k]) * 1024.0) <= abs(float(W[o, k])) * 1024.0 * 2.0 ** -22


def test_host_packers_are_reentrant():
    """Packing state is a value passed down, not file-scope: concurrent calls with different layouts (K tail on / off, plain
    tiles) give the same bytes as the same calls made one after the other (the globals could flip mid-pack)."""
    import threading
    lib, L = _lib()
    rng = np.random.default_rng(3)
    H, HP = 196, 208
    T, NC = HP // 16, (HP // 16 + 1) // 2
    W = rng.standard_normal((H, H)).astype(np.float32)

    def split(ktail):
        out = np.zeros(NC * T * 2 * 256, np.float32)
        assert lib.gaudi_host_pack_matrix_split(H, H, 0, HP, 0, ktail, 2048.0, L.fptr(W), L.fptr(out)) == 0
        return out

    def plain(tr):
        out = np.zeros(HP * HP, np.float32)
        assert lib.gaudi_host_pack_matrix(H, H, 0, HP, tr, L.fptr(W), L.fptr(out)) == 0
        return out

    jobs = [(split, 1), (split, 0), (plain, 0), (plain, 1)] * 3
    want = [f(a).tobytes() for f, a in jobs]
    got = [None] * len(jobs)

    def run(i):
        f, a = jobs[i]
        got[i] = f(a).tobytes()

    for _ in range(3):
        th = [threading.Thread(target=run, args=(i,)) for i in range(len(jobs))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert got == want
