"""CPU check of the arithmetic behind the opt-in split-operand sweeps (csrc/trmm_f16x2.hpp,
trmm_bf16x3.hpp): an f32 product rebuilt from two scaled fp16 planes (three products) or three bf16
planes (six products), accumulated in f32, is as close to the f64 result of the sweep's contraction
||Linv k*||^2 as plain f32 arithmetic is.  NumPy emulation of the planes (the GPU kernels are checked
against the oracle in tests/test_gpu_parity.py and test_gpu_configs.py)."""
import numpy as np
import pytest


def _to_bf16(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    r = ((u >> 16) & 1) + 0x7FFF
    return ((u + r) & 0xFFFF0000).view(np.float32)


def _split_bf16x3(x):
    a1 = _to_bf16(x)
    a2 = _to_bf16(x - a1)
    a3 = _to_bf16(x - a1 - a2)
    return a1, a2, a3


def _split_fp16x2(x):
    s = 2.0 ** np.floor(np.log2(16384.0 / np.max(np.abs(x))))
    xs = (x * s).astype(np.float32)
    a1 = xs.astype(np.float16).astype(np.float32)
    a2 = ((xs - a1) * 2048.0).astype(np.float16).astype(np.float32)
    return a1, a2, s


def _problem(N, M, D, noise, kind, seed):
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 1, (N, D))
    C = rng.uniform(0, 1, (M, D))
    C[::7] = X[rng.randint(0, N, len(C[::7]))]            # copies of observed points among the candidates
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
    return Linv, Ks


@pytest.mark.parametrize("N,D,noise,kind", [(384, 16, 1e-2, "rbf"), (384, 6, 1e-4, "rbf"), (300, 3, 1e-6, "matern32"),
                                            (256, 2, 1e-8, "rbf")])
def test_split_planes_reproduce_f32_products(N, D, noise, kind):
    Linv, Ks = _problem(N, 200, D, noise, kind, N + D)
    q64 = ((Linv @ Ks.T) ** 2).sum(0)
    L32, K32 = Linv.astype(np.float32), Ks.astype(np.float32)
    e32 = np.max(np.abs(((L32 @ K32.T).astype(np.float64) ** 2).sum(0) - q64))
    # two scaled fp16 planes, three products (hi and the 2^-11 pair accumulated apart)
    l1, l2, sa = _split_fp16x2(L32)
    k1, k2, sb = _split_fp16x2(K32)
    hi = l1 @ k1.T
    mid = l1 @ k2.T + l2 @ k1.T
    v = ((hi + mid * np.float32(1.0 / 2048.0)) * np.float32(1.0 / (sa * sb))).astype(np.float64)
    eh2 = np.max(np.abs((v ** 2).sum(0) - q64))
    # three bf16 planes, six products
    a1, a2, a3 = _split_bf16x3(L32)
    b1, b2, b3 = _split_bf16x3(K32)
    v6 = (a1 @ b1.T) + ((a1 @ b2.T) + (a2 @ b1.T)) + ((a1 @ b3.T) + (a2 @ b2.T) + (a3 @ b1.T))
    ex3 = np.max(np.abs((v6.astype(np.float64) ** 2).sum(0) - q64))
    # the planes carry what they claim: 22 / 24 significant bits
    assert np.max(np.abs((l1 + l2 / 2048.0) / sa - L32)) <= 2.0 ** -21 * np.max(np.abs(L32))
    np.testing.assert_array_equal((a1 + a2) + a3, L32)
    scale = max(e32, 1e-7 * (1.0 + noise))
    assert eh2 <= 2.0 * scale, (eh2, e32)
    assert ex3 <= 2.0 * scale, (ex3, e32)
