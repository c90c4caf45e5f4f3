#!/usr/bin/env python3
"""What the SHORT calls of the everyday regime cost, wall clock through the ctypes binding (round 6):

  fit_us / fit_grad_us      one tgp_fit / tgp_fit_grad (one evaluation of the hyper-parameter objective:
                            turbo/modules/surrogates.py:313-318 -> _gpr.py:584-650) at N = 8 ... 128
  acq_grad_us[m]            one tgp_acq_grad of m points (a round of the gradient stage,
                            turbo/modules/auxiliary_optimisers.py:69-112) at N = 30, 100, 900, 2048
  lbfgsb_ms, evals          tgp_fit_lbfgsb from three starts (the default construct_model's library call)

The environment selects the paths (csrc/tuning.hpp; read once per process):
  default                               polled completion, one-launch fit + gradient, one-launch query for N <= 128
  TGP_POLL_US=0 TGP_SMALL_FUSED=0 TGP_SMALL_QUERY=0    round 5's calls (events, stream synchronisation, copies)
One JSON line per size:  python tools/bench_short_calls.py [--reps 2000]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_call_us(f, reps):
    for _ in range(max(20, reps // 20)):
        f()
    best = []
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        best.append((time.perf_counter() - t0) / reps * 1e6)
    return float(min(best))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=2000)
    args = ap.parse_args()
    import turbo_amd as ta
    env = {k: os.environ[k] for k in ("TGP_POLL_US", "TGP_SMALL_FUSED", "TGP_SMALL_QUERY", "TGP_HYPER_THREADS") if k in os.environ}
    gp = ta.NativeGP(0, "f64")
    for N, D in ((8, 2), (16, 2), (32, 2), (64, 2), (100, 4), (128, 4)):
        rng = np.random.RandomState(N + D)
        X = rng.uniform(0, 1, (N, D))
        y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
        ls = float(np.sqrt(D / 6.0))
        out = {"what": "objective", "N": N, "D": D, "env": env}
        out["fit_us"] = per_call_us(lambda: gp.fit(X, y, "matern52", 1.0, ls, 1e-2, 1e-10, True), args.reps)
        out["fit_device_us"] = gp.last_timings()["fit_ms"] * 1e3
        out["fit_grad_us"] = per_call_us(lambda: gp.fit_grad(X, y, "matern52", 1.0, ls, 1e-2, 1e-10, True), args.reps)
        out["fit_grad_device_us"] = gp.last_timings()["fit_ms"] * 1e3
        out["fit_grad_phases_us"] = gp.last_timings().get("small_fit_phases_us")   # inputs staged | K tile in LDS | factored | fit done | call done
        ard = np.full(D, ls)
        out["fit_grad_ard_us"] = per_call_us(lambda: gp.fit_grad(X, y, "matern52", 1.0, ard, 1e-2, 1e-10, True), args.reps)
        theta0 = np.log(np.array([[1.0, ls, 1e-2], [0.5, 0.3, 1e-3], [3.0, 2.0, 0.1]]))
        bounds = np.log(np.array([[1e-5, 1e5]] * 3))
        res = []

        def opt():
            res.append(gp.fit_optimise(X, y, "matern52", theta0, 1, bounds, 1e-10, True, max_iter=15000, lbfgsb=True))
        opt()
        ts = []
        for _ in range(15):
            t0 = time.perf_counter()
            opt()
            ts.append(time.perf_counter() - t0)
        out["lbfgsb_ms"] = float(np.median(ts)) * 1e3
        out["lbfgsb_evals"] = int(res[-1][3])
        out["lbfgsb_us_per_eval"] = out["lbfgsb_ms"] * 1e3 / max(1, out["lbfgsb_evals"])
        print(json.dumps(out), flush=True)
    for N, D in ((30, 2), (100, 4), (900, 6), (2048, 16)):
        rng = np.random.RandomState(N)
        X = rng.uniform(0, 1, (N, D))
        y = np.sin(3 * X.sum(1)) + 0.5 * ((X - 0.5) ** 2).sum(1) + 0.01 * rng.normal(size=N)
        gp.fit(X, y, "matern52", 1.0, float(np.sqrt(D / 6.0)), 1e-3, 1e-10, True)
        out = {"what": "acq_grad", "N": N, "D": D, "env": env, "acq_grad_us": {}}
        for m in (1, 10, 64):
            P = rng.uniform(0, 1, (m, D))
            out["acq_grad_us"][str(m)] = per_call_us(lambda: gp.acq_grad(P, ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01),
                                                     args.reps if N <= 900 else args.reps // 4)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
