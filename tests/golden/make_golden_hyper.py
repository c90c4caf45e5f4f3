#!/usr/bin/env python3
"""Golden vectors for the "next" row SURVEY 8(f)1 (LML gradient + hyper-parameter optimisation),
generated FROM THE REFERENCE in the build container: the unmodified ``SciKitGPSurrogate``
(turbo/modules/surrogates.py:294-326) with ``training_iterations > 0`` and ``param_continuity``,
plus the gradient of the scikit-learn model it wraps (``log_marginal_likelihood(theta,
eval_gradient=True)``, sklearn _gpr.py:537-652).  Data only."""
import os
import sys
import warnings

import numpy as np

REF = "/root/reference"
if not os.path.isdir(os.path.join(REF, "turbo")):
    sys.exit("needs /root/reference; the committed .npz fixtures are what travels")
sys.path.insert(0, REF)
import sklearn.gaussian_process as sk_gp  # noqa: E402
import turbo.modules as tm  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
K = sk_gp.kernels


def synth(seed, N, D):
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 1, size=(N, D))
    w = rng.normal(size=D) / np.sqrt(D)
    y = np.sin(3 * X @ w) + 0.5 * ((X - 0.5) ** 2).sum(1) + 0.05 * rng.normal(size=N)
    return X, y


def grad_case(name, X, y, kernel, kind, c, ls, noise):
    sur = tm.SciKitGPSurrogate(model_params=dict(kernel=kernel, optimizer=None, normalize_y=True),
                               training_iterations=1)
    model, _ = sur.construct_model(0, X, y)
    gpr = model.model
    lml, grad = gpr.log_marginal_likelihood(gpr.kernel_.theta, eval_gradient=True)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), X=X, y=y, kind=kind, constant=c,
                        length_scale=np.atleast_1d(ls), noise=(-1.0 if noise is None else noise),
                        jitter=1e-10, theta=gpr.kernel_.theta, lml=lml, grad=grad,
                        names=np.array(model.get_hyper_param_names()))
    print("wrote", name, "lml=%.6f" % lml, "grad", grad)


def opt_case(name, X_all, y_all, kernel, kind, iters, sizes):
    """the reference's own usage: the factory persists, hyper-parameters warm-start from the
    previous trial (param_continuity), ``iterations - 1`` random restarts"""
    sur = tm.SciKitGPSurrogate(model_params=dict(kernel=kernel, normalize_y=True, random_state=0),
                               training_iterations=iters, param_continuity=True)
    out = dict(X=X_all, y=y_all, kind=kind, iters=iters, sizes=np.array(sizes))
    for t, n in enumerate(sizes):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model, info = sur.construct_model(t, X_all[:n], y_all[:n])
        out["hp_%d" % t] = model.get_hyper_params()
        out["lml_%d" % t] = model.get_log_likelihood()
        out["theta_%d" % t] = model.model.kernel_.theta
        out["bounds"] = model.model.kernel_.bounds
        Xq = X_all[-16:]
        mu, sg = model.predict(Xq, return_std_dev=True)
        out["mu_%d" % t] = mu
        out["sigma_%d" % t] = sg
        print(name, "trial", t, "n", n, "hp", out["hp_%d" % t], "lml %.6f" % out["lml_%d" % t])
    out["names"] = np.array(model.get_hyper_param_names())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)


if __name__ == "__main__":
    X, y = synth(11, 40, 3)
    grad_case("grad_rbf_iso_3d", X, y, K.ConstantKernel(1.5) * K.RBF(0.6) + K.WhiteKernel(1e-2), "rbf", 1.5, 0.6, 1e-2)
    ls = np.array([0.4, 0.7, 1.1])
    grad_case("grad_rbf_ard_3d", X, y, K.ConstantKernel(0.8) * K.RBF(ls) + K.WhiteKernel(5e-3), "rbf", 0.8, ls, 5e-3)
    X, y = synth(12, 48, 5)
    ls5 = np.linspace(0.5, 1.5, 5)
    grad_case("grad_matern52_ard_5d", X, y, K.ConstantKernel(2.0) * K.Matern(ls5, nu=2.5) + K.WhiteKernel(1e-2), "matern52", 2.0, ls5, 1e-2)
    grad_case("grad_matern32_iso_5d", X, y, K.ConstantKernel(1.0) * K.Matern(0.9, nu=1.5) + K.WhiteKernel(1e-3), "matern32", 1.0, 0.9, 1e-3)
    grad_case("grad_matern12_iso_5d_nowhite", X, y, K.ConstantKernel(1.2) * K.Matern(0.8, nu=0.5), "matern12", 1.2, 0.8, None)
    grad_case("grad_default_matern52_white", X, y, 1.0 * K.Matern(nu=2.5) + K.WhiteKernel(), "matern52", 1.0, 1.0, 1.0)
    # optimisation traces
    X, y = synth(21, 40, 2)
    opt_case("opt_default_2d", X, y, 1.0 * K.Matern(nu=2.5) + K.WhiteKernel(), "matern52", 3, [12, 20, 40])
    X, y = synth(22, 60, 4)
    opt_case("opt_rbf_ard_4d", X, y, K.ConstantKernel(1.0) * K.RBF(np.ones(4)) + K.WhiteKernel(1e-2), "rbf", 2, [30, 60])
