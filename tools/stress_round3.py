#!/usr/bin/env python3
"""Round-3 stress run (GPU): the new code paths at many shapes, against themselves under the A/B switches
(run this script under both settings of a switch and diff the digests) and against the host backend.

    python tools/stress_round3.py fit      digests of L / Linv / alpha at ~60 sizes (TGP_PANEL_FUSE=0|1 must agree)
    python tools/stress_round3.py hyper    tgp_fit_optimise at many (N, D, S): digests (TGP_HYPER_WGS=1|unset must agree)
    python tools/stress_round3.py host     GPU vs host backend on random models (prints the largest deviations)"""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turbo_amd as ta                      # noqa: E402
from turbo_amd import _lib                  # noqa: E402


def dig(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def fit():
    gp = ta.NativeGP(0, "f64")
    sizes = sorted(set([129, 191, 192, 193, 255, 256, 257, 320, 383, 384, 385, 448, 511, 512, 513, 576, 640, 767, 768, 769,
                        832, 1023, 1024, 1025, 1088, 1216, 1279, 1280, 1281, 1344, 1535, 1536, 1537, 1600, 1791, 1792,
                        1793, 2047, 2048, 2049, 2112, 2303, 2304, 2305, 2559, 2560, 2561, 3071, 3072, 3073, 3136, 3583,
                        3584, 4095, 4096, 4097, 4160, 5119, 5120, 6144, 7000]))
    if os.environ.get("TGP_STRESS_QUICK"):
        sizes = [129, 192, 257, 511, 640, 1025, 1280, 1537, 2049, 2560, 3073, 4097]
    kinds = ["rbf", "matern12", "matern32", "matern52"]
    for i, N in enumerate(sizes):
        D = 1 + (i * 7) % 19
        rng = np.random.RandomState(N)
        X = rng.uniform(0, 1, (N, D))
        y = np.sin(3 * X.sum(1)) + 0.02 * rng.normal(size=N)
        lml, _, _ = gp.fit(X, y, kinds[i % 4], 0.7 + 0.1 * (i % 5), 0.6 + 0.05 * (i % 7), 1e-3, 1e-10, bool(i % 3))
        print(json.dumps(dict(N=N, D=D, lml=repr(lml), d=dig(gp.debug_read(_lib.BUF_L), gp.debug_read(_lib.BUF_LINV),
                                                              gp.debug_read(_lib.BUF_ALPHA)))), flush=True)


def hyper():
    gp = ta.NativeGP(0, "f64")
    cases = [(65, 1, 1, False), (65, 3, 64, True), (96, 5, 7, True), (100, 2, 3, False), (127, 8, 21, True),
             (128, 4, 3, False), (128, 16, 64, True), (128, 62, 10, True), (128, 64, 5, False), (70, 33, 40, True),
             (64, 4, 64, True), (30, 2, 3, False)]
    if os.environ.get("TGP_STRESS_QUICK"):
        cases = [cases[1], cases[4], cases[6], cases[8], cases[10]]
    kinds = ["rbf", "matern52", "matern32", "matern12"]
    for i, (N, D, S, ard) in enumerate(cases):
        rng = np.random.RandomState(1000 + i)
        X = rng.uniform(0, 1, (N, D))
        y = np.sin(3 * X @ rng.normal(size=D)) + 0.5 * ((X - 0.5) ** 2).sum(1) + 0.05 * rng.normal(size=N)
        n_ls = D if ard else 1
        P = 2 + n_ls
        bounds = np.tile(np.log([1e-5, 1e5]), (P, 1))
        th0 = np.vstack([np.zeros(P)] + [rng.uniform(bounds[:, 0], bounds[:, 1]) for _ in range(S - 1)])
        theta, f, st, ev = gp.fit_optimise(X, y, kinds[i % 4], th0, n_ls, bounds, 1e-10, True, max_iter=200)
        print(json.dumps(dict(N=N, D=D, S=S, ard=ard, evals=int(ev), best=repr(float(np.nanmin(f))), status=sorted(set(st.tolist())),
                              d=dig(theta, f, st))), flush=True)


def host():
    gpu = ta.NativeGP(0, "f64")
    cpu = ta.NativeGP(_lib.DEVICE_HOST, "f64")
    worst = dict(lml=0.0, L=0.0, mu=0.0, var=0.0, acq=0.0, idx=0)
    kinds = ["rbf", "matern12", "matern32", "matern52"]
    for i in range(24):
        rng = np.random.RandomState(500 + i)
        N = int(rng.choice([5, 40, 64, 65, 128, 129, 300, 700, 1100]))
        D = int(rng.randint(1, 12))
        M = int(rng.choice([1, 17, 500, 3000]))
        X = rng.uniform(0, 1, (N, D))
        y = np.sin(3 * X.sum(1)) + 0.05 * rng.normal(size=N)
        Xc = np.vstack([rng.uniform(0, 1, (M, D)), X[:3]])
        ls = rng.uniform(0.4, 1.5, D) if i % 2 else float(rng.uniform(0.4, 1.5))
        args = (X, y, kinds[i % 4], float(rng.uniform(0.5, 2)), ls, 10.0 ** rng.uniform(-4, -2), 1e-10, bool(i % 3))
        a = gpu.fit(*args)
        b = cpu.fit(*args)
        worst["lml"] = max(worst["lml"], abs(a[0] - b[0]) / abs(b[0]))
        worst["L"] = max(worst["L"], float(np.abs(gpu.debug_read(_lib.BUF_L) - cpu.debug_read(_lib.BUF_L)).max()))
        acq = [_lib.ACQ_EI, _lib.ACQ_PI, _lib.ACQ_UCB, _lib.ACQ_SIGMA][i % 4]
        ra = gpu.evaluate(Xc, acq, -1.0, float(y.min()), 0.01 if acq != _lib.ACQ_UCB else 2.0, True, True, True)
        rb = cpu.evaluate(Xc, acq, -1.0, float(y.min()), 0.01 if acq != _lib.ACQ_UCB else 2.0, True, True, True)
        worst["mu"] = max(worst["mu"], float(np.abs(ra["mu"] - rb["mu"]).max() / max(a[2], 1e-300)))
        worst["var"] = max(worst["var"], float(np.abs(ra["sigma"] ** 2 - rb["sigma"] ** 2).max() / a[2] ** 2))
        worst["acq"] = max(worst["acq"], float(np.abs(ra["acq"] - rb["acq"]).max()))
        worst["idx"] += int(ra["best_idx"] != rb["best_idx"] and abs(ra["best_val"] - rb["best_val"]) > 1e-9 * max(1.0, abs(rb["best_val"])))
    print(json.dumps(worst))
    assert worst["lml"] < 1e-9 and worst["L"] < 1e-9 and worst["mu"] < 1e-8 and worst["var"] < 1e-8 and worst["idx"] == 0, worst


if __name__ == "__main__":
    {"fit": fit, "hyper": hyper, "host": host}[sys.argv[1]]()
