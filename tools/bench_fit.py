#!/usr/bin/env python3
"""Fit latency of tgp_fit (device time between the library's hipEvents, and host wall clock) at a
list of sizes:  python tools/bench_fit.py 512 2048 4096 [--dtype f32] [--reps 20]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sizes", type=int, nargs="+")
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--dim", type=int, default=16)
    ap.add_argument("--check", action="store_true", help="compare LML with the oracle (test infrastructure)")
    args = ap.parse_args()
    import turbo_amd as ta
    gp = ta.NativeGP(0, args.dtype)
    for N in args.sizes:
        D = args.dim
        rng = np.random.RandomState(N)
        X = rng.uniform(0, 1, (N, D))
        y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
        ls = float(np.sqrt(D / 6.0))
        dev, wall = [], []
        lml = None
        for _ in range(args.reps + 3):
            t0 = time.perf_counter()
            lml, _, _ = gp.fit(X, y, "rbf", 1.0, ls, 1e-3, 1e-10, True)
            wall.append((time.perf_counter() - t0) * 1e3)
            dev.append(gp.profile_read()["last_fit_ms"])
        out = dict(N=N, D=D, dtype=args.dtype, fit_ms_device=float(np.median(dev[3:])),
                   fit_ms_wall=float(np.median(wall[3:])), lml=lml,
                   env={k: v for k, v in os.environ.items() if k.startswith("TGP_")})
        if args.check:
            from oracle import gp_oracle as o
            om = o.fit(X, y, "rbf", 1.0, ls, 1e-3, 1e-10, True)
            out["lml_rel_err"] = abs(lml - om.lml) / abs(om.lml)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
