#!/usr/bin/env python3
"""Repeatability soak of the library's calls (GPU): every call is repeated and must return the same bytes each
time -- first alone, then with three other threads driving fits and sweeps on private streams at the same time,
so that every kernel meets foreign workgroups on its CUs.  A synchronisation that is almost always satisfied
(tools/repeat_fit.py tells the story of one) shows up as a repetition that differs from the first.

    python tools/repeat_paths.py [--reps 40] [--quick]         one JSON line per call; exit code 1 on any difference"""
import argparse
import hashlib
import json
import os
import sys
import threading

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turbo_amd as ta                      # noqa: E402
from turbo_amd import _lib                  # noqa: E402


def dig(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def data(N, D, M, seed):
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 1, (N, D))
    y = np.sin(3 * X.sum(1)) + 0.02 * rng.normal(size=N)
    return X, y, rng.uniform(0, 1, (M, D))


def calls(quick):
    """(name, function returning a digest); every function owns its handle and data"""
    out = []

    def fit_sweep(N, D, M, dtype, kind, acq):
        gp = ta.NativeGP(0, dtype)
        X, y, Xc = data(N, D, M, N + D)
        def run():
            lml, _, _ = gp.fit(X, y, kind, 1.1, 0.8, 1e-3, 1e-10, True)
            gp.set_candidates(Xc)
            r = gp.sweep(acq, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)
            return dig(np.float64(lml), r["mu"], r["sigma"], r["acq"], np.int64(r["best_idx"]))
        return run
    shapes = [(100, 5, 4000), (200, 6, 10000), (700, 7, 20000), (2304, 9, 30000), (5000, 12, 40000)]
    if quick:
        shapes = shapes[:4]
    for (N, D, M) in shapes:
        for dtype in (("f64", "f32") if N < 2304 else ("f64", "f32", "f32x3", "f32h2")):
            out.append(("fit+sweep N=%d %s" % (N, dtype), fit_sweep(N, D, M, dtype, "matern52", _lib.ACQ_EI)))

    def fit_sweep_overlapped(N, D, M, dtype, kind, acq):
        """round 5: the batch resident, tgp_set_overlap(2): the sweep's front runs inside the fit on the device's third
        stream.  Must equal the strictly serial schedule (checked once here) and itself, every repetition"""
        gp = ta.NativeGP(0, dtype)
        X, y, Xc = data(N, D, M, N + D)
        gp.fit(X, y, kind, 1.1, 0.8, 1e-3, 1e-10, True)
        gp.set_candidates(Xc)
        def run(mode=2):
            gp.set_overlap(mode)
            lml, _, _ = gp.fit(X, y, kind, 1.1, 0.8, 1e-3, 1e-10, True)
            r = gp.sweep(acq, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)
            return dig(np.float64(lml), r["mu"], r["sigma"], r["acq"], np.int64(r["best_idx"]))
        assert run(0) == run(2) == run(1), "the overlapped schedule differs from the serial one"
        return run
    for (N, D, M) in shapes[2:]:
        for dtype in ("f64", "f32"):
            out.append(("fit+sweep overlapped N=%d %s" % (N, dtype), fit_sweep_overlapped(N, D, M, dtype, "matern52", _lib.ACQ_EI)))

    def grad(N, D, ard):
        gp = ta.NativeGP(0, "f64")
        X, y, _ = data(N, D, 1, N)
        ls = np.linspace(0.6, 1.4, D) if ard else 0.9
        def run():
            lml, g = gp.fit_grad(X, y, "matern32", 1.2, ls, 1e-3, 1e-10, True)
            return dig(np.float64(lml), g)
        return run
    for N in ((300, 1100) if quick else (300, 1100, 2304, 5000)):
        out.append(("fit_grad N=%d iso" % N, grad(N, 6, False)))
        out.append(("fit_grad N=%d ard" % N, grad(N, 6, True)))

    def topk_eval(N, D, M):
        gp = ta.NativeGP(0, "f32")
        X, y, Xc = data(N, D, M, N + 1)
        def run():
            gp.fit(X, y, "rbf", 1.0, 0.9, 1e-3, 1e-10, True)
            gp.set_candidates(Xc)
            vals, idxs = gp.sweep_topk(16, _lib.ACQ_UCB, -1.0, 0.0, 2.0)
            e = gp.evaluate(Xc[:3000], _lib.ACQ_PI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)
            return dig(vals, idxs, e["mu"], e["sigma"], e["acq"])
        return run
    out.append(("topk+evaluate N=1500", topk_eval(1500, 8, 50000)))

    def hyper(N, D):
        gp = ta.NativeGP(0, "f64")
        X, y, _ = data(N, D, 1, 3 * N)
        b = np.log(np.array([[1e-2, 1e2], [1e-2, 1e2], [1e-6, 1e0]]))
        starts = np.vstack([b.mean(axis=1), [0.5, -0.5, -4.0], [-1.0, 1.0, -6.0]])
        def run():
            theta, f, st, ev = gp.fit_optimise(X, y, "matern52", starts, 1, b, 1e-10, True)
            return dig(theta, f, st, ev)
        return run
    out.append(("fit_optimise N=100", hyper(100, 4)))
    out.append(("fit_optimise N=400", hyper(400, 5)))

    def batch():
        gp = ta.NativeGP(0, "f64")
        models = []
        for i, N in enumerate((40, 90, 128, 150, 200, 256)):
            X, y, _ = data(N, 5, 1, 7 * N)
            models.append(dict(X=X, y=y, kind="matern52", constant=1.0 + 0.1 * i, length_scale=0.8, noise=1e-3, jitter=1e-10, normalize_y=True))
        _, _, Xc = data(10, 5, 5000, 11)
        def run():
            mu, sg, lml, _ = gp.predict_batch(models, Xc, True)
            return dig(mu, sg, lml)
        return run
    out.append(("predict_batch 6 models", batch()))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=40)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--shared-stream", action="store_true", help="the three threads' handles stay on the device's shared stream")
    args = ap.parse_args()
    cs = calls(args.quick)
    ref = {}
    bad = 0
    for name, fn in cs:                                   # alone
        ref[name] = fn()
        diff = sum(fn() != ref[name] for _ in range(args.reps))
        bad += diff
        print(json.dumps(dict(call=name, phase="alone", reps=args.reps, differing=diff)), flush=True)
    # ... and beside three threads of fits + sweeps on private streams
    stop = threading.Event()
    noise_bad = [0, 0, 0]
    def noise(i):
        gp = ta.NativeGP(0, ("f64", "f32", "f32h2")[i])
        gp.set_private_stream(not args.shared_stream)
        N = (1800, 3300, 4500)[i]
        X, y, Xc = data(N, 8, 60000, 100 + i)
        first = None
        while not stop.is_set():
            lml, _, _ = gp.fit(X, y, "matern52", 1.0, 0.9, 1e-3, 1e-10, True)
            gp.set_candidates(Xc)
            r = gp.sweep(_lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True)
            d = dig(np.float64(lml), r["mu"], r["sigma"], np.int64(r["best_idx"]))
            first = first or d
            noise_bad[i] += d != first
    threads = [threading.Thread(target=noise, args=(i,)) for i in range(3)]
    for t in threads:
        t.start()
    try:
        for name, fn in cs:
            diff = sum(fn() != ref[name] for _ in range(args.reps))
            bad += diff
            print(json.dumps(dict(call=name, phase="beside 3 threads", reps=args.reps, differing=diff)), flush=True)
    finally:
        stop.set()
        for t in threads:
            t.join()
    bad += sum(noise_bad)
    print(json.dumps(dict(call="the three threads' own fit + sweep loops", differing=noise_bad)), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
