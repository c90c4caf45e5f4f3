#!/usr/bin/env python3
"""What ONE evaluation of the hyper-parameter objective costs on the host side of the default path (SciPy drives
tgp_fit_grad through turbo_amd/_lib.py): wall time per call against the device time, and a cProfile of 2000 calls.

    python tools/profile_objective_call.py [N]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turbo_amd as ta   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.RandomState(N + 8)
X = rng.uniform(0, 1, (N, 8))
y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
gp = ta.NativeGP(0, "f64")
k = ta.GPKernel("matern52", 1.0, float(np.sqrt(8 / 6.0)), 1e-2)
theta = k.theta.copy()


def obj_func(theta):
    k.theta = theta
    lml, grad = gp.fit_grad(X, y, k.kind, k.constant, k.length_scale, k.noise_level, 1e-10, True)
    return -lml, -k.select_gradient(grad)


for _ in range(50):
    obj_func(theta)
t0 = time.perf_counter()
for _ in range(1000):
    obj_func(theta)
wall = (time.perf_counter() - t0) / 1000 * 1e3
dev = gp.last_timings()
print("N = %d: %.4f ms wall per evaluation; device: fit %.4f + gradient %.4f ms" % (N, wall, dev["fit_ms"], dev["grad_kinv_ms"] + dev["grad_pairwise_ms"] + dev["grad_ard_ms"]))
pr = cProfile.Profile()
pr.enable()
for _ in range(2000):
    obj_func(theta)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
