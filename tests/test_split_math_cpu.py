"""The arithmetic claim behind the fp16-pair GEMMs (gaudi_amd/csrc/w8_split.h: edge GEMMs; w8_nodes_f16.h: node GEMMs), checked
in numpy: after a power-of-two scale that brings the largest magnitude to [2^14, 2^15), an fp32 number is hi + lo with two fp16
numbers to 22 significant bits (edge form: lo in hi's units; node form: lo scaled by 2^11, which keeps the 22 bits 25 binades
down), the three piece products the kernels keep reproduce a product to 2^-21, and a K = 196 dot product accumulated from them in
fp32 is as close to float64 as a plain fp32 fma chain."""
import numpy as np


def scale_for(maxabs):
    """the kernels' rule (w8_split.h: scale_for): the exponent of the largest magnitude -> 14"""
    e = (np.asarray(maxabs, np.float32).view(np.uint32) >> 23).astype(np.int64)
    return np.ldexp(np.float32(1), np.minimum(141 - e, 126)).astype(np.float32)


def split_edge(x, s):
    y = (np.asarray(x, np.float32) * s).astype(np.float32)
    hi = y.astype(np.float16)
    lo = (y - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def split_node(x, s):
    y = (np.asarray(x, np.float32) * s).astype(np.float32)
    hi = y.astype(np.float16)
    lo = ((y - hi.astype(np.float32)) * np.float32(2048)).astype(np.float16)
    return hi, lo


def test_two_fp16_pieces_carry_22_bits():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * 10.0 ** rng.integers(-6, 6, 200000)).astype(np.float32)
    for lo_bin in (0, 8, 17):  # the column's largest entry lies 2^lo_bin above the entries looked at
        s = scale_for(np.abs(x) * np.float32(2.0 ** lo_bin))
        hi, lo = split_edge(x, s)
        rec = hi.astype(np.float64) + lo.astype(np.float64)
        rel = np.abs(rec - x.astype(np.float64) * s) / np.abs(x.astype(np.float64) * s)
        assert np.isfinite(hi.astype(np.float32)).all() and rel.max() <= 2.0 ** -22, (lo_bin, rel.max())
    # below 2^-17 of the column's largest entry the edge form degrades to fp16's absolute floor, 2^-25 of a maximum of 2^14 ...
    s = scale_for(np.abs(x) * np.float32(2.0 ** 24))
    hi, lo = split_edge(x, s)
    err = np.abs(hi.astype(np.float64) + lo.astype(np.float64) - x.astype(np.float64) * s)
    assert err.max() <= 2.0 ** -25
    # ... where the node form (lo carries its own exponent) still has its 22 bits
    hi, lo = split_node(x, s)
    rec = hi.astype(np.float64) + lo.astype(np.float64) / 2048
    assert (np.abs(rec - x.astype(np.float64) * s) / np.abs(x.astype(np.float64) * s)).max() <= 2.0 ** -22


def test_three_piece_products_match_a_product():
    rng = np.random.default_rng(1)
    a = rng.standard_normal(100000).astype(np.float32)
    b = (rng.standard_normal(100000) * 5).astype(np.float32)
    ah, al = (p.astype(np.float64) for p in split_edge(a, scale_for(np.abs(a))))
    bh, bl = (p.astype(np.float64) for p in split_edge(b, scale_for(np.abs(b))))
    three = ah * bl + al * bh + ah * bh  # every piece product is exact in fp32 (11 x 11 bits)
    exact = (a.astype(np.float64) * scale_for(np.abs(a))) * (b.astype(np.float64) * scale_for(np.abs(b)))
    rel = np.abs(three - exact) / np.abs(exact)
    assert rel.max() < 2.0 ** -21  # two representation errors of 2^-22 and the dropped al.bl
    assert np.percentile(rel, 99) < 2.0 ** -22


def test_pair_dot_product_is_as_accurate_as_an_fp32_chain():
    rng = np.random.default_rng(2)
    K, R = 196, 2000
    w = (rng.standard_normal((R, K)) / np.sqrt(K)).astype(np.float32)
    x = rng.standard_normal((R, K)).astype(np.float32)
    ref = (w.astype(np.float64) * x).sum(1)
    chain = np.zeros(R, np.float32)
    for k in range(K):  # fp32 multiply-add chain (numpy has no fma: each step rounds twice, an upper bound for the fma chain)
        chain = chain + w[:, k] * x[:, k]
    sw = scale_for(np.abs(w).max())                        # one scale for the network's weights
    sx = scale_for(np.abs(x).max(1, keepdims=True))        # one per column (here: per row of the test)
    scale = np.abs(ref).max()
    for name, split, lo_unit in (("edge", split_edge, 1.0), ("node", split_node, 1.0 / 2048)):
        wh, wl = split(w, sw)
        xh, xl = split(x, sx)
        acc0 = np.zeros(R, np.float32)  # one accumulator in the edge form; the node form keeps the cross terms apart
        acc1 = np.zeros(R, np.float32)
        for c0 in range(0, K, 32):  # one matrix instruction = 32 inputs: exact products, fp32 accumulation per instruction
            sl = slice(c0, min(c0 + 32, K))
            p_hl = (wh[:, sl].astype(np.float64) * xl[:, sl]).sum(1)
            p_lh = (wl[:, sl].astype(np.float64) * xh[:, sl]).sum(1)
            p_hh = (wh[:, sl].astype(np.float64) * xh[:, sl]).sum(1)
            if name == "edge":
                for term in (p_lh, p_hl, p_hh):  # small terms first, as in the kernel
                    acc0 = (acc0.astype(np.float64) + term).astype(np.float32)
            else:
                acc1 = (acc1.astype(np.float64) + p_hl).astype(np.float32)
                acc0 = (acc0.astype(np.float64) + p_hh).astype(np.float32)
                acc1 = (acc1.astype(np.float64) + p_lh).astype(np.float32)
        got = (acc0.astype(np.float64) + acc1.astype(np.float64) * lo_unit) / (sx[:, 0].astype(np.float64) * float(sw))
        e_pair, e_chain = np.abs(got - ref).max() / scale, np.abs(chain - ref).max() / scale
        assert e_pair < 5e-7 and e_pair <= 1.5 * e_chain + 1e-7, (name, e_pair, e_chain)
