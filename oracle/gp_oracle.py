"""CPU oracle for the GP-surrogate hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  Nothing under ``turbo_amd/`` imports it; the product path is the HIP
library behind ``include/turbogp.h`` and it fails loudly when that library is missing.

What this restates
------------------
The reference (mbway/turbo) has no arithmetic of its own on this path: its
``SciKitGPSurrogate`` (turbo/modules/surrogates.py:225-365) delegates to scikit-learn's
``GaussianProcessRegressor`` and its acquisition functions
(turbo/modules/acquisition_functions.py:80-358) add ~10 lines of NumPy + ``scipy.stats.norm``.
scikit-learn is a third-party dependency that is NOT under /root/reference and is unpinned
by the reference (requirements.txt:7 says just ``sklearn``); this restatement follows the
published algorithm of scikit-learn 1.7.2 (the version in this image):

  sklearn/gaussian_process/_gpr.py   fit :272-282, :346-364   predict :441-494   LML :584-613
  sklearn/gaussian_process/kernels.py  Sum :866/:885  Product :966/:985
        ConstantKernel :1269-1277/:1310-1314  WhiteKernel :1402/:1413-1414/:1433-1435
        RBF :1553-1565  Matern :1708-1738

Pinning
-------
The reference's own tests hold no golden vector for this path (tests/test_utils.py tests
``remap`` only, tests/test_system.py is empty), so the oracle is pinned against outputs of
the reference itself: ``tests/golden/make_golden.py`` imports the reference (plus the sklearn
it calls) in the build container, runs ``SciKitGPSurrogate`` + ``EI/PI/UCB`` +
``RandomAndQuasiNewton`` on seeded inputs and commits the inputs/outputs as ``.npz``
fixtures; ``tests/test_oracle_golden.py`` checks every function below against them.

Everything is float64 (sklearn promotes: ``X / length_scale`` with a float64 length scale).
"""
import math

import numpy as np
from scipy.linalg import cho_solve, cholesky, solve_triangular
from scipy.spatial.distance import cdist, pdist, squareform
from scipy.special import ndtr

KINDS = ("rbf", "matern12", "matern32", "matern52")


def normalise_y(y, normalize_y=True):
    """y mean / population std, exact-zero std -> 1.

    sklearn/gaussian_process/_gpr.py:272-282 and
    sklearn/preprocessing/_data.py:92-110 (_handle_zeros_in_scale).
    Entered from turbo/modules/surrogates.py:318 (``model.fit(X, y)``).
    """
    y = np.asarray(y, dtype=np.float64)
    if not normalize_y:
        return y.copy(), 0.0, 1.0
    mean = np.mean(y, axis=0)
    std = np.std(y, axis=0)
    if std == 0.0:  # _handle_zeros_in_scale, scalar branch (1-D y): exact zero only
        std = 1.0
    return (y - mean) / std, float(mean), float(std)


def _stationary(dists, kind, squared):
    """k(r) for unit-amplitude stationary kernels.

    RBF: kernels.py:1556-1565 (``exp(-0.5 * sqeuclidean)``).
    Matern nu=0.5/1.5/2.5: kernels.py:1717-1724.
    """
    if kind == "rbf":
        d2 = dists if squared else dists ** 2
        return np.exp(-0.5 * d2)
    d = np.sqrt(dists) if squared else dists
    if kind == "matern12":
        return np.exp(-d)
    if kind == "matern32":
        K = d * math.sqrt(3)
        return (1.0 + K) * np.exp(-K)
    if kind == "matern52":
        K = d * math.sqrt(5)
        return (1.0 + K + K ** 2 / 3.0) * np.exp(-K)
    raise ValueError(kind)


def kernel_matrix(X, kind, constant, length_scale, noise, jitter):
    """K = c*k(X,X) + s2*I, then ``K.diag += alpha`` (the GPR jitter).

    pdist over ``X / length_scale`` (upper triangle, direct sum of squared differences),
    squareform, diagonal forced to exactly 1 (kernels.py:1556-1560 / 1711-1738), times the
    constant (Product :966, ConstantKernel :1269-1277), plus noise on the diagonal
    (Sum :866, WhiteKernel :1402), plus the jitter (_gpr.py:346-347).
    """
    X = np.atleast_2d(np.asarray(X, dtype=np.float64))
    Xs = X / np.asarray(length_scale, dtype=np.float64)
    if kind == "rbf":
        K = squareform(_stationary(pdist(Xs, metric="sqeuclidean"), kind, True))
    else:
        K = squareform(_stationary(pdist(Xs, metric="euclidean"), kind, False))
    np.fill_diagonal(K, 1)
    K = constant * K
    K[np.diag_indices_from(K)] += noise
    K[np.diag_indices_from(K)] += jitter
    return K


def cross_kernel(Xc, X, kind, constant, length_scale):
    """K* = c*k(Xc, X); the WhiteKernel cross term is 0 (kernels.py:1413-1414).

    cdist over scaled inputs (kernels.py:1562-1565 / 1713-1716); _gpr.py:443.
    """
    ls = np.asarray(length_scale, dtype=np.float64)
    Xc = np.atleast_2d(np.asarray(Xc, dtype=np.float64))
    if kind == "rbf":
        d = cdist(Xc / ls, X / ls, metric="sqeuclidean")
        return constant * _stationary(d, kind, True)
    d = cdist(Xc / ls, X / ls, metric="euclidean")
    return constant * _stationary(d, kind, False)


class GPModel:
    """Fitted state: what sklearn keeps in X_train_, L_, alpha_, _y_train_mean/_std."""

    def __init__(self, X, kind, constant, length_scale, noise, jitter,
                 y_mean, y_std, L, alpha, lml):
        self.X = X
        self.kind = kind
        self.constant = float(constant)
        self.length_scale = np.atleast_1d(np.asarray(length_scale, dtype=np.float64))
        self.noise = float(noise)
        self.jitter = float(jitter)
        self.y_mean = y_mean
        self.y_std = y_std
        self.L = L
        self.alpha = alpha
        self.lml = lml


def fit(X, y, kind, constant, length_scale, noise=0.0, jitter=1e-10, normalize_y=True):
    """Fixed-hyper-parameter GP fit.

    turbo/modules/surrogates.py:294-326 with ``optimizer=None`` ->
    sklearn _gpr.py:272-282 (normalise), :346-347 (K), :349 (cholesky, lower; raises
    numpy.linalg.LinAlgError when not PD, :350-358), :360-364 (alpha = cho_solve),
    :584-613 (log marginal likelihood value).
    """
    assert kind in KINDS
    X = np.array(X, dtype=np.float64, copy=True)
    yn, y_mean, y_std = normalise_y(y, normalize_y)
    K = kernel_matrix(X, kind, constant, length_scale, noise, jitter)
    L = cholesky(K, lower=True, check_finite=False)
    alpha = cho_solve((L, True), yn, check_finite=False)
    lml = -0.5 * float(yn @ alpha) - float(np.log(np.diag(L)).sum()) \
        - K.shape[0] / 2 * math.log(2 * math.pi)
    return GPModel(X, kind, constant, length_scale, noise, jitter, y_mean, y_std, L, alpha, lml)


def predict(model, Xc, return_std=True, chunk=None):
    """Posterior mean and standard deviation.

    turbo/modules/surrogates.py:332-338 -> sklearn _gpr.py:443-447 (mean, de-normalise),
    :454 (V = solve_triangular(L, K*^T)), :474-475 (var = kernel_.diag - einsum; the diag
    INCLUDES the white noise: kernels.py:885,985,1310-1314,1433-1435), :479-485 (clamp
    negatives to 0), :488-494 (sigma = sqrt(var * y_std^2)).
    Rows are independent, so ``chunk`` only bounds the M x N temporaries.
    """
    Xc = np.atleast_2d(np.asarray(Xc, dtype=np.float64))
    M = Xc.shape[0]
    mus = np.empty(M)
    sig = np.empty(M) if return_std else None
    step = M if not chunk else int(chunk)
    for a in range(0, M, max(step, 1)):
        b = min(M, a + step)
        Ks = cross_kernel(Xc[a:b], model.X, model.kind, model.constant, model.length_scale)
        mus[a:b] = model.y_std * (Ks @ model.alpha) + model.y_mean
        if return_std:
            V = solve_triangular(model.L, Ks.T, lower=True, check_finite=False)
            var = np.full(b - a, model.constant + model.noise)
            var -= np.einsum("ij,ji->i", V.T, V)
            var[var < 0] = 0.0
            sig[a:b] = np.sqrt(var * model.y_std ** 2)
    return (mus, sig) if return_std else mus


def _pdf(z):
    # scipy/stats/_continuous_distns.py:356-361 (_norm_pdf)
    return np.exp(-z ** 2 / 2.0) / math.sqrt(2 * math.pi)


def acquisition(kind, mus, sigmas, desired_extremum, param, incumbent=None):
    """UCB / PI / EI over (M,) mean and std vectors.

    UCB  turbo/modules/acquisition_functions.py:147-158  (sf*mu + beta*sigma; beta=inf -> sigma)
    PI   :225-247   (mask sigma != 0; Phi(diff/sigma); 0 elsewhere)
    EI   :336-358   (diff*Phi(Z) + sigma*phi(Z); 0 where sigma == 0)
    Phi = scipy.stats.norm.cdf = scipy.special.ndtr (_continuous_distns.py:368-369).
    """
    sf = 1.0 if desired_extremum == "max" else -1.0
    mus = np.asarray(mus, dtype=np.float64)
    sigmas = np.asarray(sigmas, dtype=np.float64)
    if kind == "ucb":
        if math.isinf(param):
            return sigmas.copy()
        return sf * mus + param * sigmas
    mask = sigmas != 0
    s = sigmas[mask]
    diff = sf * (mus[mask] - incumbent) - param
    Z = diff / s
    out = np.zeros_like(mus)
    if kind == "pi":
        out[mask] = ndtr(Z)
    elif kind == "ei":
        out[mask] = diff * ndtr(Z) + s * _pdf(Z)
    else:
        raise ValueError(kind)
    return out


def sweep(model, Xc, acq_kind, desired_extremum, param, incumbent=None, chunk=None):
    """Stage 1 of the acquisition maximiser over a given candidate batch.

    turbo/modules/auxiliary_optimisers.py:59-66: ``random_y = -acq(random_x)``,
    ``argsort`` ascending, element 0 -> best; :117-129 result ``(x (1,D), max_acq)``.
    Ties are unspecified by the reference (unstable sort); the build fixes lowest index,
    which is what ``np.argmax`` returns.
    """
    mus, sig = predict(model, Xc, True, chunk=chunk)
    acq = acquisition(acq_kind, mus, sig, desired_extremum, param, incumbent)
    i = int(np.argmax(acq))
    return acq, i, float(acq[i])


def random_candidates(num_points, bounds):
    """turbo/modules/naive_selectors.py:39-46: one ``np.random.uniform`` column per
    parameter from the GLOBAL NumPy RNG, hstacked into (M, D)."""
    cols = [np.random.uniform(lo, hi, size=(num_points, 1)) for (_, lo, hi) in bounds]
    return np.hstack(cols)


# ---------------------------------------------------------------------------------------------
# "next" row SURVEY 8(f)1: log-marginal-likelihood gradient and hyper-parameter optimisation
# ---------------------------------------------------------------------------------------------

def lml_and_grad(X, y, kind, constant, length_scale, noise=None, jitter=1e-10, normalize_y=True):
    """LML and its gradient w.r.t. the LOG hyper-parameters [log c, log l (1 or D), log noise].

    sklearn _gpr.py:579-650: ``0.5 * einsum("ijl,jik->kl", alpha alpha^T - K^-1, K_gradient)``;
    kernel gradients kernels.py: ConstantKernel :1280-1290, Product :968-975, Sum :868-871,
    WhiteKernel :1403-1410, RBF :1566-1580, Matern :1740-1779.  Entered from
    turbo/modules/surrogates.py:313-318 whenever ``training_iterations > 0``.
    ``noise=None`` means no WhiteKernel term (no noise component in the gradient).
    """
    X = np.asarray(X, dtype=np.float64)
    ls = np.atleast_1d(np.asarray(length_scale, dtype=np.float64))
    yn, _, _ = normalise_y(y, normalize_y)
    s2 = 0.0 if noise is None else float(noise)
    K = kernel_matrix(X, kind, constant, length_scale, s2, jitter)
    L = cholesky(K, lower=True, check_finite=False)
    alpha = cho_solve((L, True), yn, check_finite=False)
    n = K.shape[0]
    lml = -0.5 * float(yn @ alpha) - float(np.log(np.diag(L)).sum()) - n / 2 * math.log(2 * math.pi)
    Kinv = cho_solve((L, True), np.eye(n), check_finite=False)
    G = np.outer(alpha, alpha) - Kinv

    Xs = X / ls
    aniso = ls.shape[0] > 1
    if aniso:
        Dd = (Xs[:, None, :] - Xs[None, :, :]) ** 2            # (n, n, D)
    else:
        Dd = squareform(pdist(Xs, metric="sqeuclidean"))[:, :, None]
    r2 = Dd.sum(-1)
    k0 = squareform(_stationary(pdist(Xs, metric="sqeuclidean"), kind, True)) if n > 1 else np.zeros((1, 1))
    np.fill_diagonal(k0, 1)
    if kind == "rbf":
        Kg = Dd * k0[..., None]
        Kg[np.arange(n), np.arange(n)] = 0.0   # squareform(dists) has a zero diagonal
    elif kind == "matern12":
        den = np.sqrt(r2)[:, :, None]
        div = np.zeros_like(Dd)
        np.divide(Dd, den, out=div, where=den != 0)
        Kg = k0[..., None] * div
    elif kind == "matern32":
        Kg = 3 * Dd * np.exp(-np.sqrt(3 * r2))[..., None]
    elif kind == "matern52":
        tmp = np.sqrt(5 * r2)[..., None]
        Kg = 5.0 / 3.0 * Dd * (tmp + 1) * np.exp(-tmp)
    else:
        raise ValueError(kind)
    grads = [0.5 * np.sum(G * (constant * k0))]                       # d/d log c  (Product rule)
    grads.extend(0.5 * np.einsum("ij,ijd->d", G, constant * Kg))      # d/d log l
    if noise is not None:
        grads.append(0.5 * np.trace(G) * s2)                          # d/d log noise
    return lml, np.asarray(grads, dtype=np.float64)
