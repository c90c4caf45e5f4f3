#!/usr/bin/env python3
"""Soak of the POLLED calls (round 6, csrc/doorbell.hpp): a call's results travel through device-mapped host memory and
its completion is a sequence number the last kernel stores behind a system-scope fence.  A result read before it has
landed would be an OLD value of the same buffer -- so every repetition here alternates between TWO inputs whose
results differ, and each result must equal the bytes that input gave the first time.  Tens of thousands of repetitions
per call, alone and with sixteen child processes burning CPU beside the poll loop (a descheduled poller falls back to
hipStreamSynchronize: TGP_POLL_US).

    python tools/soak_polled.py [--seconds 8]          one line per call; exit code 1 on any stale / wrong result"""
import argparse
import hashlib
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turbo_amd as ta            # noqa: E402
from turbo_amd import _lib as L   # noqa: E402


def dig(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(np.asarray(a, dtype=np.float64)).tobytes())
    return h.hexdigest()[:16]


def problem(N, D, seed):
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 1, (N, D))
    y = np.sin(3 * X.sum(1)) + 0.3 * ((X - 0.4) ** 2).sum(1) + 0.02 * rng.normal(size=N)
    return X, y


def cases():
    out = []
    for N, D in ((12, 2), (64, 3), (100, 4), (128, 6)):
        gp = ta.NativeGP(0, "f64")
        X, y = problem(N, D, N)
        ls = [0.5, 0.9]

        def fit_grad(k, gp=gp, X=X, y=y, ls=ls):
            lml, g = gp.fit_grad(X, y, "matern52", 1.0 + 0.3 * k, ls[k], 1e-2, 1e-10, True)
            return dig([lml], g)
        out.append(("fit_grad N=%d" % N, fit_grad))

        def fit(k, gp=gp, X=X, y=y, ls=ls):
            return dig(gp.fit(X, y, "rbf", 1.0 + 0.3 * k, ls[k], 1e-2, 1e-10, True))
        out.append(("fit N=%d" % N, fit))
    for N, D, m in ((30, 2, 1), (100, 4, 10), (900, 6, 10), (2048, 16, 10)):
        gp = ta.NativeGP(0, "f64")
        X, y = problem(N, D, N)
        gp.fit(X, y, "matern52", 1.0, float(np.sqrt(D / 6.0)), 1e-3, 1e-10, True)
        rng = np.random.RandomState(N)
        P = [rng.uniform(0, 1, (m, D)), rng.uniform(0, 1, (m, D))]

        def acq_grad(k, gp=gp, P=P, y=y):
            v, g = gp.acq_grad(P[k], L.ACQ_EI, -1.0, float(y.min()), 0.01)
            return dig(v, g)
        out.append(("acq_grad N=%d m=%d" % (N, m), acq_grad))
    for N, D, M in ((50, 3, 2000), (200, 4, 3000)):
        gp = ta.NativeGP(0, "f64")
        X, y = problem(N, D, N)
        gp.fit(X, y, "matern52", 1.0, float(np.sqrt(D / 6.0)), 1e-3, 1e-10, True)
        rng = np.random.RandomState(N)
        C = [rng.uniform(0, 1, (M, D)), rng.uniform(0, 1, (M, D))]

        def evaluate(k, gp=gp, C=C, y=y):
            r = gp.evaluate(C[k], L.ACQ_EI, -1.0, float(y.min()), 0.01, True, True, True)
            return dig(r["mu"], r["sigma"], r["acq"], [r["best_val"], r["best_idx"], r["n_clamped"]])
        out.append(("evaluate N=%d M=%d" % (N, M), evaluate))

        def topk(k, gp=gp, C=C, y=y):
            gp.set_candidates(C[k])
            ti, tv = gp.sweep_topk(8, L.ACQ_UCB, -1.0, 0.0, 2.0)
            return dig(ti, tv)
        out.append(("set_candidates + sweep_topk N=%d M=%d" % (N, M), topk))
    for N, D in ((300, 4), (1000, 5)):
        gp = ta.NativeGP(0, "f64")
        X, y = problem(N, D, N)
        ls = [0.5, 0.9]

        def fit_grad_blocked(k, gp=gp, X=X, y=y, ls=ls, D=D):
            lml, g = gp.fit_grad(X, y, "matern52", 0.8 + 0.3 * k, np.full(D, ls[k]), 3e-3, 1e-10, True)
            return dig([lml], g)
        out.append(("fit_grad (blocked, ARD) N=%d" % N, fit_grad_blocked))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=8.0, help="per call and per mode")
    ap.add_argument("--burners", type=int, default=16, help="child processes burning CPU in the loaded mode (a GPU box gives 16 cores)")
    args = ap.parse_args()
    bad = 0
    for mode in ("alone", "loaded"):
        burners = []
        if mode == "loaded":   # child PROCESSES spinning (threads of this interpreter would only take turns on its lock)
            burners = [subprocess.Popen([sys.executable, "-c", "while True: pass"]) for _ in range(args.burners)]
        for name, fn in cases():
            want = [fn(0), fn(1)]
            assert want[0] != want[1], name + ": the two inputs must give different results"
            n = wrong = 0
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < args.seconds:
                for k in (0, 1, 1, 0, 1, 0, 0, 1):
                    if fn(k) != want[k]:
                        wrong += 1
                    n += 1
            dt = time.perf_counter() - t0
            bad += wrong
            print("%-7s %-42s reps %7d  wrong %d  %.1f us per call" % (mode, name, n, wrong, dt / n * 1e6), flush=True)
        for b in burners:
            b.kill()
            b.wait()
    print("soak done: %d wrong" % bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
