#!/usr/bin/env python3
"""An optimised ``construct_model`` (the reference's DEFAULT usage: hyper-parameters fitted at every trial,
turbo/modules/surrogates.py:313-324; three starts) through the plugin classes, wall clock, over the sizes:
the default path (optimizer='fmin_l_bfgs_b' = tgp_fit_lbfgsb: L-BFGS-B inside the library, a C++ thread and a stream per
start) beside optimizer='scipy' (SciPy's L-BFGS-B driving the GPU objective from Python, the starts in Python threads
for 64 < N <= 1536: the default of rounds 1-4), optimizer='device' (tgp_fit_optimise: N <= 128 one launch with a
projected L-BFGS, above as the default) and, when importable, scikit-learn on the host.  One JSON line per size:

    python tools/bench_hyper_fit.py > gpurun_out/hyper_fit.jsonl
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SIZES = [(32, 2), (64, 2), (128, 4), (200, 8), (256, 8), (400, 8), (500, 8), (1000, 8), (2048, 16)]


def med(f, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def main():
    import turbo_amd as ta
    try:
        from sklearn.gaussian_process import GaussianProcessRegressor
        from sklearn.gaussian_process import kernels as K
    except ImportError:
        GaussianProcessRegressor = None
    for N, D in SIZES:
        rng = np.random.RandomState(N + D)
        X = rng.uniform(0, 1, (N, D))
        y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
        ls = float(np.sqrt(D / 6.0))
        out = {"N": N, "D": D, "starts": 3}
        for opt in ("fmin_l_bfgs_b", "scipy", "device"):
            sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, ls, 1e-2), normalize_y=True,
                                                      optimizer=opt), training_iterations=3, param_continuity=False,
                                    incremental=False)

            def fit():
                np.random.seed(11)
                return sur.construct_model(0, X, y)
            fit()
            out[opt + "_ms"] = med(fit, 7)
            m, info = fit()
            out[opt + "_lml"] = float(m.get_log_likelihood())
            out[opt + "_evals"] = info["lml_evaluations"]
            sur.close()
        if GaussianProcessRegressor is not None and N <= 1000:
            k2 = K.ConstantKernel(1.0) * K.Matern(ls, nu=2.5) + K.WhiteKernel(1e-2)

            def fit_s():
                np.random.seed(11)
                return GaussianProcessRegressor(kernel=k2, alpha=1e-10, normalize_y=True, n_restarts_optimizer=2).fit(X, y)
            out["sklearn_ms"] = med(fit_s, 1 if N >= 500 else 3)
            out["sklearn_lml"] = float(fit_s().log_marginal_likelihood_value_)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
