"""csrc/host_lbfgs.hpp on the CPU: the projected L-BFGS that tgp_fit_optimise runs above the one-launch sizes
(one C++ thread per start driving the GPU objective), compiled here behind a C entry (tests/host_lbfgs_driver.cpp)
and held against SciPy's L-BFGS-B -- the optimiser scikit-learn's GaussianProcessRegressor.fit uses
(sklearn _gpr.py:654-670, reached from turbo/modules/surrogates.py:313-318) -- on bounded test functions and on
the log marginal likelihood of the oracle."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import scipy.optimize

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CB = ctypes.CFUNCTYPE(ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), ctypes.c_void_p)
DP = ctypes.POINTER(ctypes.c_double)


@pytest.fixture(scope="module")
def minimise(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hl") / "libhost_lbfgs_test.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "turbo_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host_lbfgs_driver.cpp"), "-o", so], check=True, timeout=300)
    lib = ctypes.CDLL(so)
    lib.host_lbfgs_minimise.argtypes = [CB, ctypes.c_void_p, ctypes.c_int, DP, DP, DP, ctypes.c_int, ctypes.c_double,
                                        ctypes.c_double, DP, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]

    def run(f, x0, lo, hi, max_iter=500):
        P = len(x0)

        def cb(xp, gp, _):
            v, g = f(np.array([xp[i] for i in range(P)]))
            for i in range(P):
                gp[i] = g[i]
            return float(v)
        x = np.array(x0, dtype=np.float64)
        lo, hi = np.array(lo, dtype=np.float64), np.array(hi, dtype=np.float64)
        fo, ev, it = ctypes.c_double(), ctypes.c_int(), ctypes.c_int()
        st = lib.host_lbfgs_minimise(CB(cb), None, P, x.ctypes.data_as(DP), lo.ctypes.data_as(DP), hi.ctypes.data_as(DP),
                                     max_iter, 1e-5, 2.220446049250313e-09, ctypes.byref(fo), ctypes.byref(ev), ctypes.byref(it))
        return x, fo.value, st, ev.value
    return run


def _rosen(x):
    return scipy.optimize.rosen(x), scipy.optimize.rosen_der(x)


def test_rosenbrock_free_and_bounded(minimise):
    for P in (2, 5, 20, 66):
        x0 = np.full(P, -1.2)
        x, f, st, ev = minimise(_rosen, x0, np.full(P, -5.0), np.full(P, 5.0), max_iter=5000)
        ref = scipy.optimize.minimize(_rosen, x0, jac=True, method="L-BFGS-B", bounds=[(-5, 5)] * P)
        assert st == 1 and f <= max(ref.fun, 1e-9) * 10 + 1e-9 and ev <= 10 * ref.nfev + 60      # (a plain projected L-BFGS, memory 8: no Cauchy point, no subspace step)
        np.testing.assert_allclose(x, np.ones(P), atol=2e-3)
    # the minimiser outside the box: ends on the bounds, where SciPy ends
    lo, hi = np.array([-2.0, -2.0, -2.0]), np.array([0.5, 2.0, 0.2])
    x, f, st, ev = minimise(_rosen, np.array([-1.0, 1.0, 0.0]), lo, hi)
    ref = scipy.optimize.minimize(_rosen, np.array([-1.0, 1.0, 0.0]), jac=True, method="L-BFGS-B", bounds=list(zip(lo, hi)))
    assert st == 1 and f <= ref.fun * (1 + 1e-6) + 1e-9
    assert np.all(x >= lo) and np.all(x <= hi)
    np.testing.assert_allclose(x, ref.x, atol=1e-3)


def test_degenerate_objectives(minimise):
    # not finite at the start: gives up (status 2); a start ON the optimum: converged at once; a wall of inf beside it
    x, f, st, ev = minimise(lambda x: (np.inf, np.zeros(2)), np.zeros(2), [-1, -1], [1, 1])
    assert st == 2 and ev == 1
    x, f, st, ev = minimise(lambda x: (float(x @ x), 2 * x), np.zeros(3), [-1] * 3, [1] * 3)
    assert st == 1 and ev == 1 and f == 0.0

    def walled(x):
        if x[0] > 0.5:
            return np.inf, np.zeros(1)
        return float((x[0] - 0.45) ** 2), np.array([2 * (x[0] - 0.45)])
    x, f, st, ev = minimise(walled, np.array([-3.0]), [-4.0], [4.0])
    assert st == 1 and abs(x[0] - 0.45) < 1e-4


@pytest.mark.parametrize("kind,N,D,ard", [("rbf", 40, 2, False), ("matern52", 60, 3, True), ("matern32", 90, 4, False)])
def test_log_marginal_likelihood_of_the_oracle(minimise, kind, N, D, ard):
    """from scikit-learn's own starts it ends at SciPy's optimum (LML to 1e-6) or at a better one"""
    from oracle import gp_oracle as o
    rng = np.random.RandomState(N)
    X = rng.uniform(0, 1, (N, D))
    y = np.sin(4 * X[:, 0]) + 0.3 * X[:, -1] + 0.05 * rng.normal(size=N)
    n_ls = D if ard else 1

    def f(th):
        p = np.exp(th)
        try:
            lml, g = o.lml_and_grad(X, y, kind, p[0], p[1:1 + n_ls] if ard else p[1], p[-1], 1e-10, True)
        except np.linalg.LinAlgError:
            return np.inf, np.zeros_like(th)
        return -lml, -g
    b = np.log(np.array([[1e-5, 1e5]] * (2 + n_ls)))
    starts = [np.log(np.r_[1.0, np.full(n_ls, 0.7), 1e-2])] + [rng.uniform(b[:, 0], b[:, 1]) for _ in range(2)]
    best_ref = min(scipy.optimize.minimize(f, s, jac=True, method="L-BFGS-B", bounds=b).fun for s in starts)
    res = [minimise(f, s, b[:, 0], b[:, 1]) for s in starts]
    best = min(r[1] for r in res)
    assert all(r[2] in (1, 2) for r in res)
    assert best <= best_ref + 1e-6 * abs(best_ref)
