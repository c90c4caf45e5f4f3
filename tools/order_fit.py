#!/usr/bin/env python3
"""Fits of many sizes in random order on ONE handle against the same fits on a fresh handle each: the likelihood, alpha
and a short sweep must be bit-identical -- what an earlier, larger or smaller fit left in the buffers (the inverse
factor's zero regions, the workspace of the inverse, the candidates' slab) must not matter.

    python tools/order_fit.py [--count 60] [--max-n 6000] [--seed 1]          exit code 1 on any difference"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import turbo_amd as ta                      # noqa: E402
from turbo_amd import _lib                  # noqa: E402


def one(gp, N, dtype):
    rng = np.random.RandomState(N)
    X = rng.uniform(0, 1, (N, 6))
    y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
    Xc = rng.uniform(0, 1, (3000, 6))
    lml, _, _ = gp.fit(X, y, "matern52", 1.2, 0.9, 1e-3, 1e-10, True)
    gp.set_candidates(Xc)
    r = gp.sweep(_lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True)
    h = hashlib.sha256()
    for a in (np.float64(lml), gp.debug_read(_lib.BUF_ALPHA), r["mu"], r["sigma"], np.int64(r["best_idx"])):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--count", type=int, default=60)
    ap.add_argument("--max-n", type=int, default=6000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--dtype", default="f64")
    args = ap.parse_args()
    rng = np.random.RandomState(args.seed)
    pool = [n for n in (60, 100, 129, 200, 256, 257, 400, 512, 513, 700, 1000, 1024, 1100, 1300, 1537, 2000, 2049, 2304, 2600, 3000,
                        3073, 3700, 4096, 4097, 4500, 5000, 5500, 6000, 6700, 7000, 7800, 8192) if n <= args.max_n]
    seq = [int(pool[i]) for i in rng.randint(0, len(pool), args.count)]
    fresh = {}
    for N in sorted(set(seq)):
        gp = ta.NativeGP(0, args.dtype)
        fresh[N] = one(gp, N, args.dtype)
        gp.close()
    gp = ta.NativeGP(0, args.dtype)
    bad = []
    for i, N in enumerate(seq):
        d = one(gp, N, args.dtype)
        if d != fresh[N]:
            bad.append(dict(step=i, N=N, previous=seq[i - 1] if i else None))
    print(json.dumps(dict(dtype=args.dtype, sequence=seq, differing=bad)), flush=True)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
