#!/usr/bin/env python3
"""Accuracy of the f32 sweep and of the two opt-in split-operand sweeps (f32h2: two scaled fp16
planes, f32x3: three bf16 planes) against the f64 oracle over conditioning: one JSON line per
(N, D, kernel, noise) with the largest mean / variance error (in the test suite's units: y_std and
(c + noise) y_std^2) over 4000 candidates, a tenth of them exact copies of training points (the
variance-clamp path).  python tools/bench_split_accuracy.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import turbo_amd as ta
    from oracle import gp_oracle as o            # the checker (test infrastructure)
    for (N, D, kind, c) in ((2048, 8, "rbf", 1.0), (2048, 16, "matern52", 3.0), (1500, 4, "matern32", 0.5), (4096, 32, "rbf", 1.0)):
        for noise in (1e-2, 1e-4, 1e-6):
            rng = np.random.RandomState(N + D)
            X = rng.uniform(0, 1, (N, D))
            y = np.sin(3 * X.sum(1)) + 0.5 * ((X - 0.5) ** 2).sum(1) + 0.01 * rng.normal(size=N)
            Xc = rng.uniform(0, 1, (4000, D))
            Xc[::10] = X[rng.randint(0, N, 400)]
            ls = float(np.sqrt(D / 6.0))
            om = o.fit(X, y, kind, c, ls, noise, 1e-10, True)
            mu, sg = o.predict(om, Xc)
            out = {"N": N, "D": D, "kernel": kind, "constant": c, "noise": noise, "max_abs_Linv": float(np.abs(np.linalg.inv(om.L)).max()) if N <= 2048 else None}
            for dt in ("f32", "f32h2", "f32x3"):
                gp = ta.NativeGP(0, dt)
                gp.fit(X, y, kind, c, ls, noise, 1e-10, True)
                gp.set_candidates(Xc)
                r = gp.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True)
                out[dt] = {"mu": float(np.max(np.abs(r["mu"] - mu)) / om.y_std),
                           "var": float(np.max(np.abs(r["sigma"] ** 2 - sg ** 2)) / ((c + noise) * om.y_std ** 2)),
                           "clamped": int(r.get("n_clamped", 0))}
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
