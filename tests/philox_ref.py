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


def lhs_perm(i, n, d, seed):
    """the keyed permutation of 0..n-1 behind the device LHS: 4-round Feistel over Philox, cycle-walked"""
    bits = 2
    while (1 << bits) < n:
        bits += 1
    if bits & 1:
        bits += 1
    half = bits // 2
    mask = (1 << half) - 1
    x = np.asarray(i, dtype=np.uint64).copy()
    todo = np.ones(x.shape, dtype=bool)
    while todo.any():
        xs = x[todo]
        L = xs >> np.uint64(half)
        R = xs & np.uint64(mask)
        for r in range(4):
            F = philox4x32_10(R, r, d, 0x4C485321, seed & 0xFFFFFFFF, seed >> 32)[0] & np.uint64(mask)
            L, R = R, L ^ F
        xs = (L << np.uint64(half)) | R
        x[todo] = xs
        todo[todo] = xs >= np.uint64(n)
    return x


def lhs_design(seed, first_sample, M, n_total, lo, hi):
    """what tgp_gen_candidates_lhs / tgp_lhs_design produce: (M, D) float64"""
    lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
    D = lo.shape[0]
    out = np.empty((M, D))
    smp = np.uint64(first_sample) + np.arange(M, dtype=np.uint64)
    for d in range(D):
        pi = lhs_perm(smp, n_total, d, seed)
        e = smp * np.uint64(D) + np.uint64(d)
        draw = e >> np.uint64(1)
        r = philox4x32_10(draw & MASK, draw >> np.uint64(32), 1, 0, seed & 0xFFFFFFFF, seed >> 32)
        odd = (e & np.uint64(1)).astype(bool)
        a = np.where(odd, r[2], r[0])
        b = np.where(odd, r[3], r[1])
        u = ((a >> np.uint64(5)).astype(np.float64) * 67108864.0 + (b >> np.uint64(6)).astype(np.float64)) / 9007199254740992.0
        t = (pi.astype(np.float64) + u) / float(n_total)
        out[:, d] = lo[d] + (hi[d] - lo[d]) * t
    return out
