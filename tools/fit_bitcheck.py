#!/usr/bin/env python3
"""SHA-256 of what a fit leaves behind (L, Linv, alpha, LML) at a list of sizes, one JSON line per
size.  Run it under two settings of a TGP_* switch and diff the output: equal digests = the two
kernel paths are bit-identical (used for TGP_PANEL_FUSE=0|1).

    python tools/fit_bitcheck.py 300 700 1100 2304 4096 [--dtype f64]"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sizes", type=int, nargs="+")
    ap.add_argument("--dtype", default="f64")
    ap.add_argument("--dim", type=int, default=7)
    args = ap.parse_args()
    import turbo_amd as ta
    gp = ta.NativeGP(0, args.dtype)
    for N in args.sizes:
        D = args.dim
        rng = np.random.RandomState(N)
        X = rng.uniform(0, 1, (N, D))
        y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
        lml, _, _ = gp.fit(X, y, "matern52", 1.3, 0.9, 1e-3, 1e-10, True)
        dig = {}
        for name, which in (("L", ta._lib.BUF_L), ("Linv", ta._lib.BUF_LINV), ("alpha", ta._lib.BUF_ALPHA)):
            dig[name] = hashlib.sha256(np.ascontiguousarray(gp.debug_read(which)).tobytes()).hexdigest()[:16]
        print(json.dumps(dict(N=N, lml=repr(lml), **dig)), flush=True)


if __name__ == "__main__":
    main()
