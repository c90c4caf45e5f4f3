#!/usr/bin/env python3
"""Golden vector for the gradient stage of the acquisition maximiser, generated
from the reference's own RandomAndQuasiNewton (turbo/modules/auxiliary_optimisers.py:16-129) in the
build container.  Harness-side shims only (the reference targets 2018 libraries):
np.asscalar, and scipy.optimize.minimize given the reference's 2-D x0 flattened.  Data only."""
import os
import sys
import warnings

import numpy as np

REF = "/root/reference"
if not os.path.isdir(os.path.join(REF, "turbo")):
    sys.exit("needs /root/reference; the committed .npz fixture is what travels")
sys.path.insert(0, REF)
import scipy.optimize  # noqa: E402
import sklearn.gaussian_process as sk_gp  # noqa: E402

if not hasattr(np, "asscalar"):
    np.asscalar = lambda a: np.asarray(a).item()
_minimize = scipy.optimize.minimize


def _minimize_1d(fun, x0, *a, **k):
    f1 = lambda x: float(np.asarray(fun(x)).reshape(-1)[0])
    return _minimize(f1, np.asarray(x0).reshape(-1), *a, **k)


scipy.optimize.minimize = _minimize_1d

import turbo as tb  # noqa: E402
import turbo.modules as tm  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
K = sk_gp.kernels


def branin(x, y):
    from math import pi
    return (y - (5.1 / (4 * pi ** 2)) * x ** 2 + 5 * x / pi - 6) ** 2 + 10 * (1 - 1 / (8 * pi)) * np.cos(x) + 10


if __name__ == "__main__":
    rng = np.random.RandomState(7)
    X = np.hstack([rng.uniform(-5, 10, size=(20, 1)), rng.uniform(0, 15, size=(20, 1))])
    y = branin(X[:, 0], X[:, 1])
    out = dict(X=X, y=y)
    for name, fac, args in (("ei", tm.EI(xi=0.01), [float(y.min())]), ("ucb", tm.UCB(beta=2.0), [])):
        sur = tm.SciKitGPSurrogate(model_params=dict(kernel=2.0 * K.Matern(length_scale=3.0, nu=2.5) + K.WhiteKernel(1e-2),
                                                     optimizer=None, normalize_y=True), training_iterations=1)
        model, _ = sur.construct_model(0, X, y)
        acq, _ = fac.construct_function(0, model, 'min', *args)
        aux = tm.RandomAndQuasiNewton(num_random=256, grad_restarts=6, start_from_best=2)
        drawn = []
        gen = aux.gen_random

        def rec(n, lb):
            c = gen(n, lb)
            drawn.append(c.copy())
            return c
        aux.gen_random = rec
        np.random.seed(123)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            x, info = aux(tb.Bounds([('x', -5., 10.), ('y', 0., 15.)]), acq)
        out[name + "_x"] = x
        out[name + "_max_acq"] = info['max_acq']
        out[name + "_batch"] = drawn[0]
        out[name + "_starts"] = drawn[1]
        out[name + "_random_best"] = float(np.max(acq(drawn[0])))
        print(name, "x", x, "max_acq", info['max_acq'], "random-stage best", out[name + "_random_best"])
    np.savez_compressed(os.path.join(HERE, "stage2_branin.npz"), **out)
