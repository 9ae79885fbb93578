"""The arithmetic claim behind the split-operand edge GEMMs (gaudi_amd/csrc/w8_split.h), checked in numpy: an fp32 number is
exactly the sum of three bf16 numbers, the six piece products the kernel keeps reproduce a product to within one fp32 rounding
error, and a K = 196 dot product accumulated from them in fp32 is as close to float64 as a plain fp32 fma chain."""
import numpy as np


def bf16_rne(x):
    """round-to-nearest-even to bf16, returned as float32"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    h = bf16_rne(x)
    m = bf16_rne(x - h)
    l = bf16_rne((x - h) - m)
    return h, m, l


def test_three_bf16_pieces_are_exact():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * 10.0 ** rng.integers(-6, 6, 200000)).astype(np.float32)
    h, m, l = split3(x)
    assert np.array_equal((h.astype(np.float64) + m + l).astype(np.float32), x)
    assert np.array_equal(h.astype(np.float64) + m + l, x.astype(np.float64))  # exact, not merely rounded back
    nz = x != 0
    assert np.all(np.abs(m[nz]) <= np.abs(x[nz]) * 2.0 ** -8) and np.all(np.abs(l[nz]) <= np.abs(x[nz]) * 2.0 ** -16)


def test_six_piece_products_match_a_product_to_one_fp32_rounding():
    rng = np.random.default_rng(1)
    a = rng.standard_normal(100000).astype(np.float32)
    b = (rng.standard_normal(100000) * 5).astype(np.float32)
    ah, am, al = (p.astype(np.float64) for p in split3(a))
    bh, bm, bl = (p.astype(np.float64) for p in split3(b))
    six = ah * bh + ah * bm + am * bh + am * bm + ah * bl + al * bh  # every piece product is exact in fp32 (8 x 8 bits)
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() < 2.0 ** -23  # dropped am.bl + al.bm + al.bl; one fp32 rounding is 2^-24 relative
    assert np.percentile(rel, 99) < 2.0 ** -24


def test_split_dot_product_is_as_accurate_as_an_fp32_chain():
    rng = np.random.default_rng(2)
    K, R = 196, 2000
    w = (rng.standard_normal((R, K)) / np.sqrt(K)).astype(np.float32)
    x = rng.standard_normal((R, K)).astype(np.float32)
    ref = (w.astype(np.float64) * x).sum(1)
    chain = np.zeros(R, np.float32)
    for k in range(K):  # fp32 multiply-add chain (numpy has no fma: each step rounds twice, an upper bound for the fma chain)
        chain = chain + w[:, k] * x[:, k]
    wp, xp = split3(w), split3(x)
    order = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]  # (weight piece, input piece), smallest terms first as in the kernel
    acc = np.zeros(R, np.float32)
    for c0 in range(0, K, 32):  # one matrix instruction = 32 inputs: exact products, fp32 accumulation per instruction
        sl = slice(c0, min(c0 + 32, K))
        for i, j in order:
            acc = (acc.astype(np.float64) + (wp[i][:, sl].astype(np.float64) * xp[j][:, sl]).sum(1)).astype(np.float32)
    scale = np.abs(ref).max()
    e_split, e_chain = np.abs(acc - ref).max() / scale, np.abs(chain - ref).max() / scale
    assert e_split < 5e-7 and e_split <= 1.5 * e_chain + 1e-7, (e_split, e_chain)
