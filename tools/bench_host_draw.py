#!/usr/bin/env python3
"""The reference's host candidate draw (turbo/modules/naive_selectors.py:39-46) at BASELINE's batch shapes: NumPy's own
loop against the library's continuation of the same MT19937 stream (tgp_mt19937_uniform_columns) -- same numbers, same
RNG state afterwards (checked here too), wall ms (best of `--reps`).  One JSON line per shape.

    python tools/bench_host_draw.py [--reps 5] [--gpu] [--threads-list 1,4,16]   (TGP_HOST_THREADS is read once per process: the
    thread counts are measured in child processes)"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(reps, gpu=False):
    from turbo_amd import _lib
    for name, M, D in (("C1", 65536, 8), ("C2", 131072, 16), ("C3", 262144, 32), ("C4", 1048576, 64), ("default", 10000, 2)):
        lo = np.linspace(-1.0, 2.0, D)
        hi = lo + np.linspace(0.5, 3.0, D)
        np.random.seed(1)
        t_np = []
        for _ in range(max(1, reps if M * D < 3e7 else 1)):
            t0 = time.perf_counter()
            a = np.hstack([np.random.uniform(l, h, size=(M, 1)) for l, h in zip(lo, hi)])
            t_np.append(time.perf_counter() - t0)
            if len(t_np) == 1:
                a_first = a          # the first draw behind the seed
        after_a = np.random.uniform(size=4)
        np.random.seed(1)
        t_lib = []
        for r in range(len(t_np)):
            t0 = time.perf_counter()
            b = _lib.numpy_global_uniform_columns(M, lo, hi)
            t_lib.append(time.perf_counter() - t0)
        after_b = np.random.uniform(size=4)
        rec = {"shape": name, "M": M, "D": D, "numpy_ms": min(t_np) * 1e3, "library_ms": min(t_lib) * 1e3,
               "library_ms_all": [round(t * 1e3, 2) for t in t_lib],
               "same_numbers": bool(b is not None and np.array_equal(a, b)),
               "same_state_after": bool(np.array_equal(after_a, after_b)),
               "threads_env": os.environ.get("TGP_HOST_THREADS", ""), "host_cpus": os.cpu_count()}
        if gpu:
            # the batch made RESIDENT: NumPy's array uploaded (tgp_set_candidates) against the stream finished on the GPU
            # (tgp_set_candidates_mt19937: only the generator's recurrence on the host)
            import turbo_amd as ta
            gp = ta.NativeGP(0, "f64")
            Xt = np.random.RandomState(0).uniform(0, 1, (16, D))
            gp.fit(Xt, Xt.sum(1), "rbf", 1.0, 1.0, 1e-3, 1e-10, True)
            t_up, t_st = [], []
            for r in range(len(t_np) + 1):
                t0 = time.perf_counter()
                gp.set_candidates(a)
                t_up.append(time.perf_counter() - t0)
            np.random.seed(1)
            for r in range(len(t_np) + 1):
                t0 = time.perf_counter()
                ok = gp.set_candidates_numpy_stream(M, lo, hi)
                t_st.append(time.perf_counter() - t0)
                if r == 0:      # (the first draw after the seed is the array `a` was)
                    rec["resident_same_numbers"] = bool(ok and np.array_equal(gp.read_candidates(), a_first))
            rec["upload_ms"] = min(t_up[1:]) * 1e3
            rec["resident_stream_ms"] = min(t_st[1:]) * 1e3
            rec["numpy_plus_upload_ms"] = rec["numpy_ms"] + rec["upload_ms"]
        print(json.dumps(rec), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--threads-list", default="")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--gpu", action="store_true", help="also: the batch made resident on GPU 0, NumPy + upload against the stream finished on the GPU")
    args = ap.parse_args()
    if args.child or not args.threads_list:
        return one(args.reps, args.gpu)
    for t in args.threads_list.split(","):
        env = dict(os.environ, TGP_HOST_THREADS=t)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", "--reps", str(args.reps)] + (["--gpu"] if args.gpu else []), env=env)


if __name__ == "__main__":
    main()
