#!/usr/bin/env python3
"""What one evaluation of the hyper-parameter objective (tgp_fit_grad) costs on the caller's handle, on one pooled
worker alone, and on two and three workers side by side (a host thread each), and what a three-start hyper-parameter
fit costs through the plugin at the same sizes.  For A/B runs of the stream switches:

    TGP_BG_LEASE=0 python tools/hyper_side_by_side.py [N ...]"""
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turbo_amd as ta   # noqa: E402

SIZES = [int(a) for a in sys.argv[1:]] or [700, 1000, 1500]


def data(N):
    rng = np.random.RandomState(N + 8)
    X = rng.uniform(0, 1, (N, 8))
    return X, np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)


for N in SIZES:
    X, y = data(N)
    gp = ta.NativeGP(0, "f64")
    k = ta.GPKernel("matern52", 1.0, float(np.sqrt(8 / 6.0)), 1e-2)

    def loop(w, n):
        t0 = time.perf_counter()
        for _ in range(n):
            w.fit_grad(X, y, k.kind, k.constant, k.length_scale, k.noise_level, 1e-10, True)
        return (time.perf_counter() - t0) / n * 1e3
    loop(gp, 30)
    a = loop(gp, 200)
    with gp.workers(3) as ws:
        for w in ws:
            loop(w, 20)
        b = loop(ws[0], 200)
        res = []
        for n in (2, 3):
            out = [0.0] * n
            th = [threading.Thread(target=lambda i=i: out.__setitem__(i, loop(ws[i], 200))) for i in range(n)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            res.append(" ".join("%.3f" % v for v in out))
    print("N = %d, ms per evaluation: caller's handle %.3f | one worker alone %.3f | two side by side %s | three %s" % (N, a, b, res[0], res[1]), flush=True)
for N in SIZES:
    X, y = data(N)
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, float(np.sqrt(8 / 6.0)), 1e-2), normalize_y=True),
                            training_iterations=3, param_continuity=False, incremental=False)
    ts = []
    for r in range(6):
        np.random.seed(11)
        t0 = time.perf_counter()
        m, info = sur.construct_model(0, X, y)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("hyper-parameter fit, three starts, N = %d: %.2f ms (%d evaluations, LML %.6f)" % (N, float(np.median(ts[1:])), info["lml_evaluations"], m.get_log_likelihood()), flush=True)
    sur.close()
