"""csrc/host_lbfgsb.hpp on the CPU: L-BFGS-B restated in C++ -- what tgp_fit_lbfgsb (and tgp_fit_optimise above the
one-launch sizes) walks every start with, one C++ thread per start driving the GPU objective -- compiled here behind
a C entry (tests/host_lbfgs_driver.cpp) and held against SciPy's L-BFGS-B, the optimiser scikit-learn's
GaussianProcessRegressor.fit uses (sklearn _gpr.py:654-670, reached from turbo/modules/surrogates.py:313-318): on
bounded test functions and on the log marginal likelihood of the oracle it must evaluate the objective at the points
SciPy evaluates it at -- same number of evaluations, same number of iterations, same optimum."""
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

    def run(f, x0, lo, hi, max_iter=15000, trace=None):
        P = len(x0)

        def cb(xp, gp, _):
            xx = np.array([xp[i] for i in range(P)])
            if trace is not None:
                trace.append(xx)
            v, g = f(xx)
            for i in range(P):
                gp[i] = g[i]
            return float(v)
        x = np.array(x0, dtype=np.float64)
        lo, hi = np.array(lo, dtype=np.float64), np.array(hi, dtype=np.float64)
        fo, ev, it = ctypes.c_double(), ctypes.c_int(), ctypes.c_int()
        st = lib.host_lbfgs_minimise(CB(cb), None, P, x.ctypes.data_as(DP), lo.ctypes.data_as(DP), hi.ctypes.data_as(DP),
                                     max_iter, 1e-5, 2.220446049250313e-09, ctypes.byref(fo), ctypes.byref(ev), ctypes.byref(it))
        return x, fo.value, st, ev.value, it.value
    return run


def _same_walk_as_scipy(minimise, f, x0, lo, hi, trace_tol=1e-7):
    """both optimisers from x0: equal evaluation and iteration counts, every evaluated point equal to trace_tol, the
    same end point.  (A search that ends on "no further progress" asks for its best point once more: SciPy answers
    from its cache of the last evaluation and the driver does the same, so the counts are of distinct points.)"""
    pts_ref, pts = [], []

    def f_ref(x):
        pts_ref.append(x.copy())
        return f(x)
    ref = scipy.optimize.minimize(f_ref, x0, jac=True, method="L-BFGS-B", bounds=list(zip(lo, hi)))
    x, fv, st, ev, it = minimise(f, x0, lo, hi, trace=pts)
    assert (ev, it) == (ref.nfev, ref.nit), (ev, it, ref.nfev, ref.nit)
    assert st == (1 if ref.status == 0 else 2)
    dev = [float(np.max(np.abs(a - b))) for a, b in zip(pts, pts_ref)]
    assert max(dev[:25]) <= 1e-7 and max(dev) <= trace_tol, (max(dev[:25]), max(dev))
    np.testing.assert_allclose(x, ref.x, atol=trace_tol)
    assert abs(fv - ref.fun) <= 1e-9 * max(1.0, abs(ref.fun))
    return ref


def _rosen(x):
    return scipy.optimize.rosen(x), scipy.optimize.rosen_der(x)


def test_rosenbrock_free_and_bounded(minimise):
    for P in (2, 5, 20, 66):
        ref = _same_walk_as_scipy(minimise, _rosen, np.full(P, -1.2), np.full(P, -5.0), np.full(P, 5.0))
        np.testing.assert_allclose(ref.x, np.ones(P), atol=2e-3)
    # the minimiser outside the box: the Cauchy point and the subspace step's projection both at work
    _same_walk_as_scipy(minimise, _rosen, np.array([-1.0, 1.0, 0.0]), np.array([-2.0, -2.0, -2.0]), np.array([0.5, 2.0, 0.2]))
    _same_walk_as_scipy(minimise, _rosen, np.zeros(4), np.array([-0.5, -0.5, 0.3, -1.0]), np.array([0.8, 0.6, 0.5, 0.1]))
    # a start outside the box is clipped into it first, as SciPy does; one coordinate with lo == hi never moves
    _same_walk_as_scipy(minimise, _rosen, np.array([3.0, -4.0, 0.5]), np.array([-1.0, -1.0, 0.5]), np.array([2.0, 2.0, 0.5]))
    # half-open and open coordinates
    _same_walk_as_scipy(minimise, _rosen, np.array([-1.2, 1.0, -0.5]), np.array([-np.inf, 0.2, -np.inf]), np.array([0.7, np.inf, np.inf]))
    _same_walk_as_scipy(minimise, _rosen, np.full(6, -1.2), np.full(6, -np.inf), np.full(6, np.inf))


def test_iteration_limit_and_quadratics(minimise):
    x, f, st, ev, it = minimise(_rosen, np.full(10, -1.2), np.full(10, -5.0), np.full(10, 5.0), max_iter=7)
    ref = scipy.optimize.minimize(_rosen, np.full(10, -1.2), jac=True, method="L-BFGS-B", bounds=[(-5, 5)] * 10, options=dict(maxiter=7))
    assert st == 0 and it == 7 and ev == ref.nfev
    np.testing.assert_allclose(x, ref.x, atol=1e-9)
    rng = np.random.RandomState(4)
    for P in (3, 12, 40):        # ill-conditioned convex quadratics, the minimiser partly outside the box
        A = rng.normal(size=(P, P))
        A = A @ A.T + 1e-3 * np.eye(P)
        b = rng.normal(size=P) * 3

        def q(x):
            return 0.5 * x @ A @ x - b @ x, A @ x - b
        _same_walk_as_scipy(minimise, q, rng.uniform(-1, 1, P), np.full(P, -1.0), np.full(P, 1.5), trace_tol=1e-6)


def test_degenerate_objectives(minimise):
    # +inf with a zero gradient at the start (K not positive definite there): a stationary point to SciPy, and here
    x, f, st, ev, it = minimise(lambda x: (np.inf, np.zeros(2)), np.zeros(2), [-1, -1], [1, 1])
    assert st == 1 and ev == 1 and f == np.inf
    x, f, st, ev, it = minimise(lambda x: (float(x @ x), 2 * x), np.zeros(3), [-1] * 3, [1] * 3)
    assert st == 1 and ev == 1 and f == 0.0

    # a wall of +inf (a kernel matrix that is not positive definite, _gpr.py:586-589): SciPy lets the inf run through the
    # line search's interpolation, which sends the search back to its best step so far -- a start whose first trial hits
    # the wall stops where it stands.  scikit-learn's restarts end that way, so must these.
    def walled(x):
        if x[0] > 0.5:
            return np.inf, np.zeros_like(x)
        return float(np.sum((x - 0.45) ** 2)), 2 * (x - 0.45)
    for x0 in ([-3.0, -1.0], [0.0, 0.0], [-0.2, 0.3], [0.44, -2.0], [0.3, 0.1, -0.1]):
        P = len(x0)
        ref = _same_walk_as_scipy(minimise, walled, np.array(x0), np.full(P, -4.0), np.full(P, 4.0))
        assert ref.nit >= 1

    def walled_later(x):         # the wall met in a later search, beside a curved valley
        if x[1] > 1.5:
            return np.inf, np.zeros_like(x)
        return _rosen(x)
    for x0 in ([-1.2, 1.0], [0.5, 1.4], [-1.5, 1.45], [1.2, 1.3]):
        _same_walk_as_scipy(minimise, walled_later, np.array(x0), np.full(2, -2.0), np.full(2, 2.0))


@pytest.mark.parametrize("kind,N,D,ard", [("rbf", 40, 2, False), ("matern52", 60, 3, True), ("matern32", 90, 4, False),
                                          ("matern12", 50, 2, True)])
def test_log_marginal_likelihood_of_the_oracle(minimise, kind, N, D, ard):
    """from scikit-learn's own starts (the kernel's theta, then uniform draws inside the bounds) it walks SciPy's
    iterates on the oracle's negative log marginal likelihood"""
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
    for s0 in starts:
        _same_walk_as_scipy(minimise, f, s0, b[:, 0], b[:, 1], trace_tol=2e-2)    # (a flat plateau amplifies rounding late in a long walk)


def test_lbfgsb_under_sanitizers(tmp_path):
    """AddressSanitizer + UBSan over csrc/host_lbfgsb.hpp (tests/host_lbfgsb_sanitizer_driver.cpp): memory wrap-around,
    active bounds, the projected subspace step, walls of inf, lo == hi, no bounds, the iteration limit"""
    exe = str(tmp_path / "lbfgsb_san")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-I" + os.path.join(ROOT, "turbo_amd", "csrc"),
           os.path.join(ROOT, "tests", "host_lbfgsb_sanitizer_driver.cpp"), "-o", exe]
    built = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if built.returncode != 0 and ("asan" in built.stderr.lower() or "sanitize" in built.stderr.lower()):
        pytest.skip("this g++ has no sanitizer runtime: " + built.stderr[-200:])
    assert built.returncode == 0, built.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "ERROR" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-3000:]
    lines = run.stdout.strip().splitlines()
    assert len(lines) == 8 and all("inside=1" in ln for ln in lines), run.stdout
    st = [int(ln.split("status=")[1].split()[0]) for ln in lines]
    assert st[:6] == [1] * 6 and st[6] == 0 and st[7] == 1, run.stdout        # converged ... | stopped by max_iter | nothing to move
    assert "iters=7 " in lines[6]


@pytest.mark.parametrize("name,kind,ls0,noise0,fixed", [
    ("opt_default_2d", "matern52", 1.0, 1.0, None), ("opt_rbf_ard_4d", "rbf", np.ones(4), 1e-2, None),
    ("opt_matern32_iso_5d_mid", "matern32", 0.9, 1e-2, None), ("opt_fixed_noise_3d", "matern52", 0.8, 1e-2, "noise"),
    ("opt_fixed_constant_ard_3d", "rbf", np.ones(3), 1e-2, "constant"), ("opt_nowhite_matern52_3d", "matern52", 0.8, None, "no noise term")])
def test_the_references_first_trial_is_reproduced_by_the_oracle_and_this_optimiser(minimise, name, kind, ls0, noise0, fixed):
    """tests/golden/opt_*.npz are outputs of the reference (SciKitGPSurrogate.construct_model with training_iterations
    > 0: scikit-learn's fit, SciPy's L-BFGS-B from the kernel's theta and from restarts drawn with random_state = 0).
    Trial 0 of every trace, on the CPU: the oracle's log marginal likelihood as the objective, this optimiser from the
    same starts (a fixed hyper-parameter left out of the vector, as scikit-learn leaves it out of theta) -- the
    reference's likelihood and hyper-parameters."""
    from conftest import golden_path
    from oracle import gp_oracle as o
    with np.load(golden_path(name), allow_pickle=False) as z:
        t = {k: z[k] for k in z.files}
    n = int(t["sizes"][0])
    X, y = t["X"][:n], t["y"][:n]
    n_ls = np.size(ls0)
    with np.errstate(divide="ignore"):
        full0 = np.log(np.r_[1.0, np.atleast_1d(ls0), 0.0 if noise0 is None else noise0])
    free = np.ones(2 + n_ls, dtype=bool)
    if fixed in ("noise", "no noise term"):
        free[-1] = False
    elif fixed == "constant":
        free[0] = False
    jitter = float(t["alpha"]) if "alpha" in t else 1e-10
    b = t["bounds"]
    assert b.shape == (int(free.sum()), 2)

    def f(th):
        full = full0.copy()
        full[free] = th
        p = np.exp(full)
        try:
            lml, g = o.lml_and_grad(X, y, kind, p[0], p[1:1 + n_ls] if n_ls > 1 else p[1], None if noise0 is None else p[-1], jitter, True)
            if noise0 is None:
                g = np.r_[g, 0.0]
        except np.linalg.LinAlgError:
            return np.inf, np.zeros_like(th)
        return -lml, -g[free]
    rng = np.random.RandomState(0)
    starts = [full0[free]] + [rng.uniform(b[:, 0], b[:, 1]) for _ in range(int(t["iters"]) - 1)]
    res = [minimise(f, s0, b[:, 0], b[:, 1]) for s0 in starts]
    best = min(res, key=lambda r: r[1])
    assert -best[1] == pytest.approx(float(t["lml_0"]), rel=1e-6, abs=1e-6)
    want = t["hp_0"]
    inner = (want > 1.1e-5) & (want < 0.9e5)
    np.testing.assert_allclose(best[0][inner], np.log(want[inner]), atol=2e-3)
