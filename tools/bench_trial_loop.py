#!/usr/bin/env python3
"""Latency of ONE Bayesian-optimisation trial in the reference's everyday regime (its Branin-Hoo
demo: 2D, tens of trials), through the plugin classes of this package on one GPU, beside the same
steps on the host's CPUs the way the reference takes them:

  surrogate   construct_model with the hyper-parameters optimised (training_iterations = 3: the
              warm start + 2 random restarts; turbo/modules/surrogates.py:294-326 ->
              GaussianProcessRegressor.fit)
  selection   RandomAndQuasiNewton (turbo/modules/auxiliary_optimisers.py:48-129): 10 000 random
              candidates -> EI -> the 2 best + 8 random starts refined by L-BFGS-B

Three stacks, one JSON line per trial count N:
  gpu_default  HipGPSurrogate() + RandomAndQuasiNewton(): both stages L-BFGS-B inside the library (tgp_fit_lbfgsb,
               tgp_acq_lbfgsb: SciPy's walk, no interpreter between evaluations)
  gpu_device   HipGPSurrogate(optimizer='device') + RandomAndQuasiNewton(on_device=True): the one-launch optimisers
               of the library's own (tgp_fit_optimise, tgp_acq_refine)
  gpu_scipy    HipGPSurrogate(optimizer='scipy') + RandomAndQuasiNewton(lockstep='scipy'): SciPy drives the GPU
               objectives from Python threads (the defaults of rounds 2-4)
  cpu          scikit-learn's GaussianProcessRegressor(n_restarts_optimizer=2) and SciPy's L-BFGS-B
               over 1-point acquisition calls with finite-difference gradients: a restatement of
               what the reference executes per trial (the reference itself does not travel to
               the GPU box), its own random stream
Every stack sees the same training points (a fixed Branin-Hoo sample); ms are medians of 5 trials.

    python tools/bench_trial_loop.py [--sizes 8 16 32 64 128]"""
import argparse
import json
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def branin(X):
    x1, x2 = X[:, 0], X[:, 1]
    a, b, c, r, s, t = 1.0, 5.1 / (4 * np.pi ** 2), 5 / np.pi, 6.0, 10.0, 1 / (8 * np.pi)
    return a * (x2 - b * x1 ** 2 + c * x1 - r) ** 2 + s * (1 - t) * np.cos(x1) + s


def med(f, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", type=int, nargs="+", default=[8, 16, 32, 64, 128])
    ap.add_argument("--num-random", type=int, default=10000)
    ap.add_argument("--restarts", type=int, default=10)
    args = ap.parse_args()
    import turbo_amd as ta
    try:
        from sklearn.gaussian_process import GaussianProcessRegressor, kernels as K
        import scipy.optimize
        from scipy.stats import norm
    except ImportError:
        GaussianProcessRegressor = None
    lo, hi = np.array([-5.0, 0.0]), np.array([10.0, 15.0])
    bounds = ta.Bounds([("x1", lo[0], hi[0]), ("x2", lo[1], hi[1])])
    warnings.simplefilter("ignore")
    for N in args.sizes:
        rng = np.random.RandomState(100 + N)
        X = rng.uniform(lo, hi, (N, 2))
        y = branin(X)
        out = {"N": N, "D": 2, "num_random": args.num_random, "restarts": args.restarts}
        for name, opt in (("gpu_default", "fmin_l_bfgs_b"), ("gpu_device", "device"), ("gpu_scipy", "scipy")):
            sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 1.0, 1.0), normalize_y=True,
                                                      random_state=0, optimizer=opt),
                                    training_iterations=3, param_continuity=False, incremental=False)
            aux = ta.RandomAndQuasiNewton(num_random=args.num_random, grad_restarts=args.restarts, start_from_best=2,
                                          on_device=(name == "gpu_device"), lockstep="scipy" if name == "gpu_scipy" else True)
            state = {}

            def fit():
                state["model"] = sur.construct_model(0, X, y)[0]

            def select():
                f, _ = ta.EI(xi=0.01).construct_function(0, state["model"], "min", float(y.min()))
                np.random.seed(5)
                state["x"], state["info"] = aux(bounds, f)

            fit(); select()
            out[name + "_fit_ms"] = med(fit)
            out[name + "_select_ms"] = med(select)
            out[name + "_trial_ms"] = out[name + "_fit_ms"] + out[name + "_select_ms"]
            out[name + "_lml"] = float(state["model"].get_log_likelihood())
            out[name + "_max_acq"] = float(state["info"]["max_acq"])
        if GaussianProcessRegressor is not None:
            kern = K.ConstantKernel(1.0) * K.Matern(1.0, nu=2.5) + K.WhiteKernel(1.0)
            state = {}

            def fit_c():
                state["g"] = GaussianProcessRegressor(kernel=kern, alpha=1e-10, normalize_y=True,
                                                      n_restarts_optimizer=2, random_state=0).fit(X, y)

            def ei(P):
                mu, sd = state["g"].predict(P, return_std=True)
                diff = -(mu - y.min()) - 0.01
                with np.errstate(divide="ignore", invalid="ignore"):
                    Z = diff / sd
                    v = diff * norm.cdf(Z) + sd * norm.pdf(Z)
                return np.where(sd != 0, v, 0.0)

            def select_c():
                np.random.seed(5)
                rx = np.random.uniform(lo, hi, (args.num_random, 2))
                ry = -ei(rx)
                ids = np.argsort(ry)
                best_y = ry[ids[0]]
                starts = np.vstack([rx[ids[:2]], np.random.uniform(lo, hi, (args.restarts - 2, 2))])
                for j in range(args.restarts):
                    res = scipy.optimize.minimize(lambda x: -ei(x[None, :])[0], starts[j], bounds=list(zip(lo, hi)),
                                                  method="L-BFGS-B", options=dict(maxiter=15000))
                    if res.success and res.fun < best_y:
                        best_y = res.fun
                state["max_acq"] = -float(best_y)

            fit_c(); select_c()
            out["cpu_fit_ms"] = med(fit_c, 3)
            out["cpu_select_ms"] = med(select_c, 3)
            out["cpu_trial_ms"] = out["cpu_fit_ms"] + out["cpu_select_ms"]
            out["cpu_lml"] = float(state["g"].log_marginal_likelihood_value_)
            out["cpu_max_acq"] = state["max_acq"]
            out["host_cpus"] = os.cpu_count()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
