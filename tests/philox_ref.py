"""NumPy Philox-4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11) used
to check the device-side candidate generator bit for bit.  Test helper, not product code."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) & MASK for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)) & MASK
        n1 = p1 & MASK
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)) & MASK
        n3 = p0 & MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def uniform_candidates(seed, first_candidate, M, lo, hi):
    """what tgp_gen_candidates produces: (M, D) float64"""
    lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
    D = lo.shape[0]
    e = np.uint64(first_candidate * D) + np.arange(M * D, dtype=np.uint64)
    draw = e >> np.uint64(1)
    r = philox4x32_10(draw & MASK, draw >> np.uint64(32), 0, 0, seed & 0xFFFFFFFF, seed >> 32)
    odd = (e & np.uint64(1)).astype(bool)
    a = np.where(odd, r[2], r[0])
    b = np.where(odd, r[3], r[1])
    u = ((a >> np.uint64(5)).astype(np.float64) * 67108864.0 + (b >> np.uint64(6)).astype(np.float64)) / 9007199254740992.0
    d = (np.arange(M * D) % D)
    return (lo[d] + (hi[d] - lo[d]) * u).reshape(M, D)
