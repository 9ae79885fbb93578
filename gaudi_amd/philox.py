"""Host twin (numpy) of the device Philox4x32-10 + Box-Muller stream (csrc/device_common.h):
key = seed, counter = (element quad, draw, global sample lo, hi).  Used to reproduce production-mode
noise on the host, e.g. to feed the same draws to another implementation."""
from __future__ import annotations

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, np.uint32) for c in (c0, c1, c2, c3))
    k0 = np.asarray(k0, np.uint32)
    k1 = np.asarray(k1, np.uint32)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c1 ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c3 ^ k1
            n3 = p0.astype(np.uint32)
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = (k0 + W0).astype(np.uint32)
            k1 = (k1 + W1).astype(np.uint32)
    return c0, c1, c2, c3


def philox_normal(seed, sample_offset, B, n_elem, draw0, n_draws) -> np.ndarray:
    """-> float32 [n_draws, B, n_elem], identical (to ~1e-6) to gaudi_philox_normal."""
    nq = (n_elem + 3) // 4
    dr, b, q = np.meshgrid(np.arange(n_draws) + draw0, np.arange(B) + sample_offset, np.arange(nq), indexing="ij")
    sample = b.astype(np.uint64)
    r = philox4x32_10(q, dr, sample & np.uint64(0xFFFFFFFF), sample >> np.uint64(32),
                      np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    f = np.float32
    inv = f(1.0 / 16777216.0)
    u1 = ((r[0] >> np.uint32(8)).astype(f) + f(1.0)) * inv
    u2 = (r[1] >> np.uint32(8)).astype(f) * inv
    u3 = ((r[2] >> np.uint32(8)).astype(f) + f(1.0)) * inv
    u4 = (r[3] >> np.uint32(8)).astype(f) * inv
    ra = np.sqrt(f(-2.0) * np.log(u1)).astype(f)
    rb = np.sqrt(f(-2.0) * np.log(u3)).astype(f)
    two_pi = f(6.283185307179586)
    out = np.stack([ra * np.cos(two_pi * u2), ra * np.sin(two_pi * u2), rb * np.cos(two_pi * u4),
                    rb * np.sin(two_pi * u4)], axis=-1).astype(f)
    return out.reshape(n_draws, B, nq * 4)[:, :, :n_elem]
