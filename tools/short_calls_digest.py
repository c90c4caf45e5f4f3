#!/usr/bin/env python3
"""The BYTES the short calls return, for comparing kernel selections of the same build (round 6): small fit,
fit + LML gradient (iso and ARD), the small-problem sweeps (evaluate, arg-max only, top-k), acquisition value +
gradient (EI / PI / UCB, the small-problem and the general kernels), the one-launch hyper-parameter fit and the library's L-BFGS-B -- one line per call with a SHA-256 of the
returned arrays and their leading values.  Two processes under different TGP_* settings (csrc/tuning.hpp is read once
per process) must print the same lines where the switch promises the same bytes:

    python tools/short_calls_digest.py > a.txt
    TGP_POLL_US=0 TGP_SMALL_FUSED=0 TGP_SMALL_LIVE=0 python tools/short_calls_digest.py > b.txt && diff a.txt b.txt

`--skip` leaves out groups a switch is NOT expected to keep bit-identical (TGP_SMALL_QUERY changes the kernels of
tgp_acq_grad at N <= 128: same values to rounding, other bytes): names of groups, comma separated."""
import argparse
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(np.asarray(a, dtype=np.float64)).tobytes())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip", default="")
    args = ap.parse_args()
    skip = set(filter(None, args.skip.split(",")))
    import turbo_amd as ta
    L = ta._lib
    gp = ta.NativeGP(0, "f64")
    for N, D, kind in ((5, 2, "matern52"), (12, 2, "matern52"), (16, 3, "rbf"), (17, 2, "matern32"), (33, 4, "matern52"),
                       (48, 2, "rbf"), (64, 5, "matern52"), (65, 3, "matern12"), (100, 4, "matern52"), (128, 8, "rbf")):
        rng = np.random.RandomState(100 * N + D)
        X = rng.uniform(0, 1, (N, D))
        y = np.sin(3 * X.sum(1)) + 0.3 * ((X - 0.4) ** 2).sum(1) + 0.02 * rng.normal(size=N)
        ls = float(np.sqrt(D / 6.0))
        ard = ls * (0.6 + 0.8 * np.arange(D) / max(D - 1, 1))
        if "fit" not in skip:
            lml, ym, ys = gp.fit(X, y, kind, 1.3, ls, 2e-2, 1e-10, True)
            print("fit      N=%3d %-8s %s lml=%.15g L=%s Linv=%s alpha=%s" % (
                N, kind, digest([lml, ym, ys]), lml, digest(np.tril(gp.debug_read(L.BUF_L))), digest(gp.debug_read(L.BUF_LINV)),
                digest(gp.debug_read(L.BUF_ALPHA))))
        if "fit_grad" not in skip:
            lml, g = gp.fit_grad(X, y, kind, 1.3, ls, 2e-2, 1e-10, True)
            print("fit_grad N=%3d %-8s %s lml=%.15g g0=%.15g" % (N, kind, digest([lml], g), lml, g[0]))
            lml, g = gp.fit_grad(X, y, kind, 0.7, ard, 5e-3, 1e-10, True)
            print("fit_gard N=%3d %-8s %s lml=%.15g g1=%.15g Linv=%s" % (N, kind, digest([lml], g), lml, g[1], digest(gp.debug_read(L.BUF_LINV))))
        P = rng.uniform(0, 1, (7, D))
        P[0] = X[0]                       # an observed point: the variance cancels
        gp.fit(X, y, kind, 1.3, ls, 2e-2, 1e-10, True)
        if "acq_grad" not in skip:
            for acq, par in ((L.ACQ_EI, 0.01), (L.ACQ_PI, 0.01), (L.ACQ_UCB, 2.0)):
                v, g = gp.acq_grad(P, acq, -1.0, float(y.min()), par)
                v1, g1 = gp.acq_grad(P[3:4], acq, -1.0, float(y.min()), par)      # a point's value does not depend on its batch
                assert v1[0] == v[3] and np.array_equal(g1[0], g[3]), (N, acq, v1, v[3])
                print("acq_grad N=%3d %-8s acq=%d %s v=%s" % (N, kind, acq, digest(v, g), np.array2string(v[:3], precision=15)))
        if "sweep" not in skip:
            C = rng.uniform(0, 1, (1500, D))
            C[7] = X[1]
            r = gp.evaluate(C, L.ACQ_EI, -1.0, float(y.min()), 0.01, True, True, True)
            gp.set_candidates(C)
            r2 = gp.sweep(L.ACQ_PI, -1.0, float(y.min()), 0.01)
            ti, tv = gp.sweep_topk(8, L.ACQ_UCB, -1.0, 0.0, 2.0)
            print("sweep    N=%3d %-8s %s best=%d/%.15g pi=%d/%.15g topk=%s" % (
                N, kind, digest(r["mu"], r["sigma"], r["acq"], [r["best_val"], r["best_idx"], r["n_clamped"]], [r2["best_val"], r2["best_idx"]], tv, ti),
                r["best_idx"], r["best_val"], r2["best_idx"], r2["best_val"], list(map(int, ti[:3]))))
        theta0 = np.log(np.array([[1.0, ls, 1e-2], [0.5, 0.3, 1e-3], [3.0, 2.0, 0.1]]))
        bounds = np.log(np.array([[1e-5, 1e5]] * 3))
        if "lbfgsb" not in skip:
            th, f, st, ev = gp.fit_optimise(X, y, kind, theta0, 1, bounds, 1e-10, True, max_iter=15000, lbfgsb=True)
            print("lbfgsb   N=%3d %-8s %s evals=%d f=%s" % (N, kind, digest(th, f, st), ev, np.array2string(f, precision=12)))
        if "device" not in skip:
            th, f, st, ev = gp.fit_optimise(X, y, kind, theta0, 1, bounds, 1e-10, True, max_iter=500, lbfgsb=False)
            print("device   N=%3d %-8s %s evals=%d f=%s st=%s" % (N, kind, digest(th, f, st), ev, np.array2string(f, precision=12), st))
    # the general query kernels (N > 128): polled against copied
    for N, D in ((200, 3), (900, 6)):
        rng = np.random.RandomState(N)
        X = rng.uniform(0, 1, (N, D))
        y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
        if "fit_grad" not in skip:      # the blocked fit + gradient: a polled chain of launches (a ring kernel behind it)
            lml, g = gp.fit_grad(X, y, "matern52", 0.8, np.full(D, 0.9), 3e-3, 1e-10, True)
            print("fit_gard N=%3d blocked  %s lml=%.15g g1=%.15g Linv=%s" % (N, digest([lml], g), lml, g[1], digest(gp.debug_read(L.BUF_LINV))))
        lml, ym, ys = gp.fit(X, y, "matern52", 1.0, float(np.sqrt(D / 6.0)), 1e-3, 1e-10, True)
        if "fit" not in skip:
            print("fit      N=%3d blocked  %s lml=%.15g L=%s Linv=%s alpha=%s" % (
                N, digest([lml, ym, ys]), lml, digest(np.tril(gp.debug_read(L.BUF_L))), digest(gp.debug_read(L.BUF_LINV)),
                digest(gp.debug_read(L.BUF_ALPHA))))
        P = rng.uniform(0, 1, (70, D))
        if "sweep" not in skip and N <= 256:      # the one-launch sweep of 128 < N <= 256: top-k polled, evaluate synchronised
            C = rng.uniform(0, 1, (3000, D))
            r = gp.evaluate(C, L.ACQ_EI, -1.0, float(y.min()), 0.01, True, True, True)
            gp.set_candidates(C)
            ti, tv = gp.sweep_topk(8, L.ACQ_EI, -1.0, float(y.min()), 0.01)
            print("sweep    N=%3d mid      %s best=%d topk=%s" % (N, digest(r["mu"], r["sigma"], r["acq"], tv, ti), r["best_idx"], list(map(int, ti[:3]))))
        if "acq_grad_general" not in skip:
            for acq, par in ((L.ACQ_EI, 0.01), (L.ACQ_UCB, 2.0)):
                v, g = gp.acq_grad(P, acq, -1.0, float(y.min()), par)
                v1, g1 = gp.acq_grad(P[5:6], acq, -1.0, float(y.min()), par)
                assert v1[0] == v[5] and np.array_equal(g1[0], g[5])
                print("acq_grad N=%3d general  acq=%d %s v=%s" % (N, acq, digest(v, g), np.array2string(v[:3], precision=15)))
    print("digest done")


if __name__ == "__main__":
    main()
