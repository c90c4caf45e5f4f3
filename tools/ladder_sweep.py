#!/usr/bin/env python3
"""Sweeps over the first M of one candidate set for a ladder of M around every chunk / group / launch boundary: mean,
deviation and acquisition of candidate i must not depend on how many candidates follow it (bit for bit), and the
reported winner is the arg-max of the vector.

    python tools/ladder_sweep.py [--dtype f32] [--n 1100 4096]          exit code 1 on any difference"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turbo_amd as ta                      # noqa: E402
from turbo_amd import _lib                  # noqa: E402

LADDER = [1, 2, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 1000, 4095, 4096, 4097, 8191, 8192, 8193, 16383, 16384,
          16385, 32767, 32768, 32769, 49153, 65535, 65536, 65537, 98305, 131071, 131072, 131073, 196609, 262143, 262144, 262145,
          270001]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", nargs="+", default=["f64", "f32", "f32x3", "f32h2"])
    ap.add_argument("--n", type=int, nargs="+", default=[100, 200, 300, 1100, 4096])
    args = ap.parse_args()
    bad = 0
    for dtype in args.dtype:
        for N in args.n:
            D = 5 if N < 1000 else 12
            rng = np.random.RandomState(N)
            X = rng.uniform(0, 1, (N, D))
            y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
            Xc = rng.uniform(0, 1, (LADDER[-1], D))
            gp = ta.NativeGP(0, dtype)
            gp.fit(X, y, "matern52", 1.2, 0.9, 1e-3, 1e-10, True)
            gp.set_candidates(Xc)
            full = gp.sweep(_lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)
            full = {k: np.array(v, copy=True) if isinstance(v, np.ndarray) else v for k, v in full.items()}
            diffs = []
            for M in LADDER:
                gp.set_candidates(Xc[:M])
                r = gp.sweep(_lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)
                ok = all(np.array_equal(r[k], full[k][:M]) for k in ("mu", "sigma", "acq")) and r["best_idx"] == int(np.argmax(r["acq"]))
                if not ok:
                    diffs.append(M)
            bad += len(diffs)
            print(json.dumps(dict(dtype=dtype, N=N, ladder=len(LADDER), differing_M=diffs)), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
