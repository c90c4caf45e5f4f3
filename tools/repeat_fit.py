#!/usr/bin/env python3
"""The same fit many times on one handle: every repetition must return the same log-likelihood, bit for bit.
A fit is a few hundred launches on two streams; a synchronisation that is almost always satisfied shows up
here as a handful of different values among hundreds (round 4: the trailing update's k-loop let a direct-to-LDS
write overtake a pending LDS read when the background stream's kernels shared its CUs -- one fit in ten at
N = 5000; TGP_GEMM64=round4-war brings that loop back to show it).

    python tools/repeat_fit.py 3700 5000 5500 [--trials 300]        exit code 1 if any size disagrees with itself"""
import argparse
import collections
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def repeat(gp, N, trials, dim=7):
    rng = np.random.RandomState(N)
    X = rng.uniform(0, 1, (N, dim))
    y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
    vals = [gp.fit(X, y, "matern52", 1.3, 0.9, 1e-3, 1e-10, True)[0] for _ in range(trials)]
    cnt = collections.Counter(vals)
    ref, nref = cnt.most_common(1)[0]
    return dict(N=N, trials=trials, lml=repr(ref), disagreeing=trials - nref, distinct=len(cnt))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sizes", type=int, nargs="+")
    ap.add_argument("--trials", type=int, default=300)
    ap.add_argument("--dtype", default="f64")
    args = ap.parse_args()
    import turbo_amd as ta
    gp = ta.NativeGP(0, args.dtype)
    bad = 0
    for N in args.sizes:
        r = repeat(gp, N, args.trials)
        r["env"] = {k: v for k, v in os.environ.items() if k.startswith("TGP_")}
        bad += r["disagreeing"]
        print(json.dumps(r), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
