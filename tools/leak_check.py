#!/usr/bin/env python3
"""Create -> use -> destroy in a loop: device memory (hipMemGetInfo through torch), host RSS, open file descriptors and
thread count before and after must not grow with the number of cycles.

    python tools/leak_check.py [--cycles 150]          exit code 1 if anything grows by more than a cycle's worth"""
import argparse
import json
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turbo_amd as ta                      # noqa: E402
from turbo_amd import _lib                  # noqa: E402


def state():
    import torch
    free, total = torch.cuda.mem_get_info(0)
    rss = int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE")
    return dict(dev_used_mb=(total - free) / 2**20, rss_mb=rss / 2**20, fds=len(os.listdir("/proc/self/fd")), threads=threading.active_count(),
                os_threads=len(os.listdir("/proc/self/task")))


def cycle(i):
    rng = np.random.RandomState(i)
    N = (90, 200, 700, 1500)[i % 4]
    X = rng.uniform(0, 1, (N, 5))
    y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
    Xc = rng.uniform(0, 1, (5000, 5))
    gp = ta.NativeGP(0, ("f64", "f32", "f32h2", "f32x3")[i % 4])
    if i % 3 == 0:
        gp.set_private_stream(True)
    gp.fit(X, y, "matern52", 1.0, 0.9, 1e-3, 1e-10, True)
    gp.set_candidates(Xc)
    gp.sweep(_lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True)
    gp.fit_grad(X, y, "matern52", 1.0, 0.9, 1e-3, 1e-10, True)
    if i % 5 == 0:
        b = np.log(np.array([[1e-2, 1e2], [1e-2, 1e2], [1e-6, 1e0]]))
        gp.fit_optimise(X[:300], y[:300], "rbf", np.vstack([b.mean(axis=1), [0.5, -0.5, -4.0]]), 1, b, 1e-10, True, max_iter=15)
    if i % 7 == 0:
        sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 1.0, 1e-2), normalize_y=True),
                                training_iterations=2, param_continuity=False, incremental=False)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sur.construct_model(0, X[:260], y[:260])
        sur.close()
    try:
        gp.fit(X, -y, "bogus", 1.0, 0.9, 1e-3, 1e-10, True)           # an error path per cycle
    except Exception:
        pass
    gp.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cycles", type=int, default=150)
    args = ap.parse_args()
    for i in range(12):
        cycle(i)                                                        # warm-up: pools, code objects, the stream pair
    keep = ta.NativeGP(0, "f64")                                        # (keeps the device's stream pair alive across cycles)
    a = state()
    for i in range(args.cycles):
        cycle(i)
    mid = state()
    for i in range(args.cycles):
        cycle(i)
    b = state()
    keep.close()
    print(json.dumps(dict(cycles=args.cycles, before=a, middle=mid, after=b)), flush=True)
    grow = {k: b[k] - mid[k] for k in b}
    bad = grow["dev_used_mb"] > 64 or grow["rss_mb"] > 64 or grow["fds"] > 4 or grow["os_threads"] > 4
    print(json.dumps(dict(growth_second_half=grow, leak=bad)), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
