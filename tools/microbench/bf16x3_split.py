#!/usr/bin/env python3
"""Numerical experiment behind DESIGN.md section 8's "fp32 contraction from three bf16 planes" lever
(NOT part of the product path): split every f32 operand into three bf16 planes a = a1 + a2 + a3 and
form a.b from the six products a1b1, a1b2, a2b1, a1b3, a2b2, a3b1 accumulated in f32 -- what six
v_mfma_f32_32x32x16_bf16 would do in 192 cycles per 32x32x16 block against 512 for the eight
v_mfma_f32_32x32x2_f32 of the present kernel.  Prints the error of q = ||Linv k*||^2 (the sweep's
contraction) against f64 for plain f32, the 6-product split and the cheaper 3-product split."""
import numpy as np


def to_bf16(x):
    """round-to-nearest-even f32 -> bf16, returned as f32"""
    u = x.astype(np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def split3(x):
    a1 = to_bf16(x)
    a2 = to_bf16(x - a1)
    a3 = to_bf16(x - a1 - a2)
    return a1, a2, a3


def main():
    rng = np.random.RandomState(0)
    N, M, D = 1024, 512, 32
    X = rng.uniform(0, 1, (N, D)); C = rng.uniform(0, 1, (M, D))
    ls = np.sqrt(D / 6.0)
    d2 = ((X[:, None, :] - X[None, :, :]) ** 2).sum(-1) / ls ** 2
    K = np.exp(-0.5 * d2) + 1e-2 * np.eye(N)
    Linv = np.linalg.inv(np.linalg.cholesky(K))
    Ks = np.exp(-0.5 * ((C[:, None, :] - X[None, :, :]) ** 2).sum(-1) / ls ** 2)      # (M, N)
    q64 = ((Linv @ Ks.T) ** 2).sum(0)
    L32, K32 = Linv.astype(np.float32), Ks.astype(np.float32)
    q32 = ((L32 @ K32.T).astype(np.float64) ** 2).sum(0)
    l1, l2, l3 = split3(L32); k1, k2, k3 = split3(K32)
    f = lambda a, b: (a @ b.T)                      # f32 x f32 -> f32 accumulate (numpy sgemm)
    v6 = f(l1, k1) + (f(l1, k2) + f(l2, k1)) + (f(l1, k3) + f(l2, k2) + f(l3, k1))
    v3 = f(l1, k1) + (f(l1, k2) + f(l2, k1))
    q6 = (v6.astype(np.float64) ** 2).sum(0); q3 = (v3.astype(np.float64) ** 2).sum(0)
    kss = 1.0 + 1e-2
    for name, q in (("f32", q32), ("bf16 x 3 planes, 6 products", q6), ("bf16 x 2 planes, 3 products", q3)):
        print("%-30s max |q - q64| / (c + s2) = %.2e" % (name, np.max(np.abs(q - q64)) / kss))


if __name__ == "__main__":
    main()
