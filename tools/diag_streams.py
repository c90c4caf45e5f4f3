#!/usr/bin/env python3
"""Which of this process's handles run their launch chains side by side?  One handle on the device's shared main
stream and three on private streams (what a threaded hyper-parameter fit uses) each repeat an LML + gradient
evaluation (a chain of ~35 dependent launches at N = 500), alone and then in groups from one host thread each.  Side by
side an evaluation costs what it costs alone (0.35 ms); two handles that the runtime serialises cost the sum (0.8 ms
each).  Round 4 found the shared-stream handle and ONE worker serialised in the first factory of a fresh process --
invisible to a single-launch overlap probe -- which is why the library's threaded paths run on workers only.

    python tools/diag_streams.py [--n 500]"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turbo_amd as ta   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=500)
    ap.add_argument("--reps", type=int, default=60)
    args = ap.parse_args()
    N, D = args.n, 8
    rng = np.random.RandomState(N + D)
    X = rng.uniform(0, 1, (N, D))
    y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
    shared = ta.NativeGP(0, "f64")
    ws = [ta.NativeGP(0, "f64") for _ in range(3)]
    for w in ws:
        w.set_private_stream(True)
    names = ["shared"] + ["private%d" % i for i in range(3)]
    hs = [shared] + ws

    def loop(g, n):
        t0 = time.perf_counter()
        for _ in range(n):
            g.fit_grad(X, y, "matern52", 1.0, 1.1, 1e-2, 1e-10, True)
        return (time.perf_counter() - t0) / n * 1e3

    for g in hs:
        loop(g, 5)
    print(json.dumps(dict(alone_ms_per_evaluation={n: round(loop(g, args.reps), 3) for n, g in zip(names, hs)})), flush=True)

    def together(idx):
        out = {}

        def run(i):
            out[names[i]] = round(loop(hs[i], args.reps), 3)
        th = [threading.Thread(target=run, args=(i,)) for i in idx]
        for t in th:
            t.start()
        for t in th:
            t.join()
        return out

    worst = 0.0
    for idx in ([0, 1], [0, 2], [0, 3], [1, 2], [1, 3], [2, 3], [1, 2, 3], [0, 1, 2], [0, 2, 3]):
        r = together(idx)
        worst = max(worst, max(r.values()))
        print(json.dumps(dict(together=r)), flush=True)
    print(json.dumps(dict(worst_ms_per_evaluation=worst)), flush=True)


if __name__ == "__main__":
    main()
