#!/usr/bin/env python3
"""Numerical experiment (NOT product code): the sweep's contraction from TWO fp16 planes per f32
operand, a*s = a1 + 2^-11 a2 (s a power of two that puts max|a| near 2^14), three products
a1 b1 + 2^-11 (a1 b2 + a2 b1) [+ 2^-22 a2 b2] accumulated in f32 -- half the MFMA work of the
three-bf16-plane scheme (tools/microbench/bf16x3_split.py) and 4 bytes per element.
Prints the error of q = ||Linv k*||^2 against f64 for several problem shapes."""
import numpy as np


def split_fp16(x):
    m = np.max(np.abs(x))
    s = 2.0 ** np.floor(np.log2(16384.0 / m))
    xs = (x * s).astype(np.float32)
    a1 = xs.astype(np.float16).astype(np.float32)
    a2 = ((xs - a1) * 2048.0).astype(np.float16).astype(np.float32)
    return a1, a2, s


def run(N, M, D, noise, kind, seed):
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 1, (N, D)); C = rng.uniform(0, 1, (M, D))
    ls = np.sqrt(D / 6.0)
    def kern(A, B):
        d2 = ((A[:, None, :] - B[None, :, :]) ** 2).sum(-1) / ls ** 2
        if kind == "rbf":
            return np.exp(-0.5 * d2)
        r = np.sqrt(3.0 * d2)
        return (1 + r) * np.exp(-r)
    K = kern(X, X) + noise * np.eye(N)
    Linv = np.linalg.inv(np.linalg.cholesky(K))
    Ks = kern(C, X)
    q64 = ((Linv @ Ks.T) ** 2).sum(0)
    L32, K32 = Linv.astype(np.float32), Ks.astype(np.float32)
    q32 = ((L32 @ K32.T).astype(np.float64) ** 2).sum(0)
    l1, l2, sa = split_fp16(L32); k1, k2, sb = split_fp16(K32)
    f = lambda a, b: (a @ b.T)
    hi = f(l1, k1); mid = f(l1, k2) + f(l2, k1); lo = f(l2, k2)
    v3 = (hi.astype(np.float64) + mid.astype(np.float64) / 2048.0) / (sa * sb)
    v4 = v3 + lo.astype(np.float64) / 2048.0 ** 2 / (sa * sb)
    # the same combined in f32, as a kernel epilogue would
    v3f = ((hi + mid * np.float32(1.0 / 2048.0)) * np.float32(1.0 / (sa * sb))).astype(np.float64)
    kss = 1.0 + noise
    e = lambda q: np.max(np.abs(q - q64)) / kss
    print("N=%4d D=%2d %-8s noise %.0e  max|Linv| %.1e:  f32 %.2e | fp16x2 3 products %.2e (combined in f32 %.2e) | 4 products %.2e"
          % (N, D, kind, noise, np.abs(Linv).max(), e(q32), e((v3 ** 2).sum(0)), e((v3f ** 2).sum(0)), e((v4 ** 2).sum(0))))


if __name__ == "__main__":
    run(1024, 512, 32, 1e-2, "rbf", 0)
    run(1024, 512, 8, 1e-4, "rbf", 1)
    run(1024, 512, 4, 1e-6, "matern32", 2)
    run(512, 512, 2, 1e-8, "rbf", 3)
