#!/usr/bin/env python3
"""Where the wall time of one optimised construct_model goes (default path: SciPy's L-BFGS-B driving the GPU objective,
three starts on worker handles): cProfile of a few fits at N = 500.

    python tools/profile_hyper_fit.py [N]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turbo_amd as ta   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
rng = np.random.RandomState(N + 8)
X = rng.uniform(0, 1, (N, 8))
y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, float(np.sqrt(8 / 6.0)), 1e-2), normalize_y=True),
                        training_iterations=3, param_continuity=False, incremental=False)


def fit():
    np.random.seed(11)
    return sur.construct_model(0, X, y)


for _ in range(3):
    fit()
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    m, info = fit()
    ts.append((time.perf_counter() - t0) * 1e3)
print("N = %d: construct_model %.2f ms median, %d evaluations, final fit %.3f ms" % (N, float(np.median(ts)), info["lml_evaluations"], info["fit_ms"]))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    fit()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
