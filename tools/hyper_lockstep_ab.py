#!/usr/bin/env python3
"""tgp_fit_lbfgsb above N = 128: the starts in lock-step through ONE chain of launches (default, round 6) against a
thread and a chain per start (TGP_HYPER_LOCKSTEP=0, round 5) -- the BYTES they return and what they cost.

One line per case: SHA-256 of (theta, -lml, status), the evaluation count, wall ms (median of `--reps` calls).  Two
processes under the two settings must print the same digests and counts (csrc/tuning.hpp is read once per process):

    python tools/hyper_lockstep_ab.py --sizes 200,500,1000 > a.txt
    TGP_HYPER_LOCKSTEP=0 python tools/hyper_lockstep_ab.py --sizes 200,500,1000 > b.txt
    diff <(cut -d' ' -f1-6 a.txt) <(cut -d' ' -f1-6 b.txt)

(the reference's call: turbo/modules/surrogates.py:313-324 -> sklearn _gpr.py:296-337, n_restarts_optimizer + 1 starts)"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(np.asarray(a, dtype=np.float64)).tobytes())
    return h.hexdigest()[:16]


def problem(N, D, seed):
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 1, (N, D))
    y = np.sin(3 * X.sum(1)) + 0.5 * ((X - 0.4) ** 2).sum(1) + 0.05 * rng.normal(size=N)
    return X, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="200,500,1000")
    ap.add_argument("--starts", type=int, default=3)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dims", type=int, default=4)
    ap.add_argument("--ard", type=int, default=1, help="also an ARD case per size")
    args = ap.parse_args()
    import turbo_amd as ta
    gp = ta.NativeGP(0, "f64")
    D = args.dims
    for N in [int(v) for v in args.sizes.split(",")]:
        X, y = problem(N, D, N)
        for n_ls in ([1, D] if args.ard else [1]):
            P = 2 + n_ls
            rng = np.random.RandomState(7 * N + n_ls)
            theta0 = np.empty((args.starts, P))
            theta0[0] = np.log([1.0] + [0.6] * n_ls + [1e-2])
            for s in range(1, args.starts):   # log-uniform in the bounds, as scikit-learn draws its restarts
                theta0[s] = rng.uniform(np.log(1e-2), np.log(1e2), P)
            bounds = np.log(np.array([[1e-5, 1e5]] * P))
            ts = []
            out = None
            for r in range(args.reps + 1):
                t0 = time.perf_counter()
                out = gp.fit_optimise(X, y, "matern52", theta0, n_ls, bounds, 1e-10, True, max_iter=15000, lbfgsb=True)
                if r:
                    ts.append(time.perf_counter() - t0)
            th, f, st, ev = out
            print("N=%d D=%d n_ls=%d S=%d %s evals=%d  ms=%.3f us_per_round=%.1f f=%s" % (
                N, D, n_ls, args.starts, digest(th, f, st), ev, np.median(ts) * 1e3,
                np.median(ts) * 1e6 / max(1, ev) * args.starts, np.array2string(f, precision=10)), flush=True)
    print("done")


if __name__ == "__main__":
    main()
