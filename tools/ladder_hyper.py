#!/usr/bin/env python3
"""tgp_fit_optimise above the one-launch sizes (a C++ thread + stream per start inside the library) over a ladder of
N, D and start counts: digests of (theta, f, status, evaluations).  Run under TGP_HYPER_THREADS=1 and unset and diff:
the outcome must not depend on how many starts run side by side.  Also repeats every case three times in-process.

    python tools/ladder_hyper.py > a.txt;  TGP_HYPER_THREADS=1 python tools/ladder_hyper.py > b.txt;  diff a.txt b.txt"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turbo_amd as ta                      # noqa: E402


def dig(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def main():
    gp = ta.NativeGP(0, "f64")
    cases = [(129, 2, 3, False), (150, 5, 4, True), (200, 8, 3, False), (256, 3, 5, True), (257, 4, 3, False), (400, 8, 3, True),
             (513, 6, 2, False), (640, 4, 4, False), (641, 4, 3, True), (1000, 8, 3, False), (1281, 5, 2, False), (1500, 6, 3, True),
             (100, 70, 3, True), (300, 66, 2, True)]
    kinds = ["rbf", "matern52", "matern32", "matern12"]
    bad = 0
    for i, (N, D, S, ard) in enumerate(cases):
        rng = np.random.RandomState(2000 + i)
        X = rng.uniform(0, 1, (N, D))
        y = np.sin(3 * X @ rng.normal(size=D)) + 0.5 * ((X - 0.5) ** 2).sum(1) + 0.05 * rng.normal(size=N)
        n_ls = D if ard else 1
        P = 2 + n_ls
        bounds = np.tile(np.log([1e-5, 1e5]), (P, 1))
        th0 = np.vstack([np.zeros(P)] + [rng.uniform(-2, 2, P) for _ in range(S - 1)])
        ds = []
        for rep in range(3):
            theta, f, st, ev = gp.fit_optimise(X, y, kinds[i % 4], th0, n_ls, bounds, 1e-10, True, max_iter=60)
            ds.append(dig(theta, f, st, np.int64(ev)))
        bad += len(set(ds)) != 1
        print(json.dumps(dict(N=N, D=D, S=S, ard=ard, evals=int(np.sum(ev)), best=repr(float(np.nanmin(f))), status=sorted(set(st.tolist())),
                              d=ds[0], repeats_equal=len(set(ds)) == 1)), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
