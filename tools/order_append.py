#!/usr/bin/env python3
"""Random walks of full fits and one-row appends (tgp_fit_append, the Optimiser's per-trial update) on ONE handle: after
every step the likelihood and a short sweep must agree with a from-scratch fit of the same rows on a fresh handle
(appended factors agree to rounding, not to the bit).  Exercises what earlier, larger fits and appends leave behind in
the buffers (the inverse factor is only zero-filled when needed) across the small / one-launch / general size classes.

    python tools/order_append.py [--steps 400] [--seed 3]          exit code 1 on any disagreement"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turbo_amd as ta                      # noqa: E402
from turbo_amd import _lib                  # noqa: E402

KERN = ("matern52", 1.2, 0.8, 2e-3)


def check(gp, fresh, lml, X, y, Xc, n, tag, bad, tol_mu):
    lf, _, _ = fresh.fit(X[:n], y[:n], *KERN, 1e-10, True)
    gp.set_candidates(Xc)
    fresh.set_candidates(Xc)
    a = gp.sweep(_lib.ACQ_EI, -1.0, float(y[:n].min()), 0.01, want_mu=True, want_sigma=True)
    b = fresh.sweep(_lib.ACQ_EI, -1.0, float(y[:n].min()), 0.01, want_mu=True, want_sigma=True)
    ok = abs(lml - lf) <= 1e-9 * max(1.0, abs(lf)) and np.allclose(a["mu"], b["mu"], rtol=tol_mu, atol=tol_mu) \
        and np.allclose(a["sigma"] ** 2, b["sigma"] ** 2, rtol=10 * tol_mu, atol=tol_mu)
    if not ok:
        bad.append(dict(step=tag, n=n, lml=(lml, lf), dmu=float(np.abs(a["mu"] - b["mu"]).max()),
                        dvar=float(np.abs(a["sigma"] ** 2 - b["sigma"] ** 2).max())))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--dtype", default="f64")
    args = ap.parse_args()
    rng = np.random.RandomState(args.seed)
    NMAX = 1400
    X = rng.uniform(0, 1, (NMAX, 5))
    y = np.sin(3 * X.sum(1)) + 0.5 * ((X - 0.5) ** 2).sum(1) + 0.02 * rng.normal(size=NMAX)
    Xc = rng.uniform(0, 1, (1500, 5))
    gp, fresh = ta.NativeGP(0, args.dtype), None
    tol = 1e-8 if args.dtype == "f64" else 5e-3
    bad, log = [], []
    n = 0
    lml = 0.0
    for step in range(args.steps):
        r = rng.rand()
        if n == 0 or r < 0.25:                       # a full fit at a size of its own
            n = int(rng.choice([20, 60, 100, 127, 128, 129, 200, 250, 256, 257, 300, 500, 511, 513, 700, 1023, 1025, 1200]))
            lml = gp.fit(X[:n], y[:n], *KERN, 1e-10, True, append=True)[0]
            log.append(("fit", n))
        else:                                        # a run of appends
            k = int(rng.randint(1, 9))
            for _ in range(k):
                if n + 1 > NMAX:
                    break
                n += 1
                lml = gp.fit(X[:n], y[:n], *KERN, 1e-10, True, append=True)[0]
            log.append(("append", k, n))
        fresh = ta.NativeGP(0, args.dtype)
        check(gp, fresh, lml, X, y, Xc, n, step, bad, tol)
        fresh.close()
    print(json.dumps(dict(dtype=args.dtype, steps=args.steps, disagreeing=bad[:10], n_bad=len(bad), walk=log[:40])), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
