#!/usr/bin/env python3
"""Latency of the auxiliary optimiser's gradient stage (reference
turbo/modules/auxiliary_optimisers.py:69-112) on one GPU, one JSON line per (N, D, acquisition):
  acq_grad_ms[b]      one batched tgp_acq_grad call (value + gradient at b points)
  topk_ms             tgp_sweep_topk: the k best of a swept batch of M candidates, on the device
  refine_ms           tgp_acq_refine alone from `restarts` starts (2 best of the batch + random ones):
                      N <= 128 one launch, a workgroup per restart; above, all restarts in lock-step
  stage_device_ms     the whole stage: candidates (NumPy) + sweep + top-k + tgp_acq_refine
  lbfgsb_ms           tgp_acq_lbfgsb alone from the same starts: L-BFGS-B per restart inside the library, lock-step
  stage_lbfgsb_ms     the whole stage with it (CandidateSweep's default)
  stage_scipy_ms      the same stage with SciPy's L-BFGS-B driving batched tgp_acq_grad calls in lock-step from Python threads
                      (lockstep='scipy': the default of rounds 2-4)
and both optima.  python tools/bench_gradient_stage.py [--restarts 10] [--num-random 10000]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def med(f, reps=7):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--restarts", type=int, default=10)
    ap.add_argument("--num-random", type=int, default=10000)
    args = ap.parse_args()
    import turbo_amd as ta
    for N, D in ((30, 2), (100, 4), (900, 6), (2048, 16)):
        rng = np.random.RandomState(N)
        X = rng.uniform(0, 1, (N, D))
        y = np.sin(3 * X.sum(1)) + 0.5 * ((X - 0.5) ** 2).sum(1) + 0.01 * rng.normal(size=N)
        b = ta.Bounds([("x%d" % d, 0.0, 1.0) for d in range(D)])
        sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, float(np.sqrt(D / 6.0)), 1e-3),
                                                  optimizer=None, normalize_y=True), training_iterations=1)
        model, _ = sur.construct_model(0, X, y)
        for name, fac, fa in (("ei", ta.EI(xi=0.01), [float(y.min())]), ("ucb", ta.UCB(beta=2.0), [])):
            f, _ = fac.construct_function(0, model, "min", *fa)
            out = {"N": N, "D": D, "acq": name, "restarts": args.restarts, "num_random": args.num_random, "acq_grad_ms": {}}
            for nb in (1, 10, 64):
                P = rng.uniform(0, 1, (nb, D))
                f.value_and_grad(P)
                out["acq_grad_ms"][str(nb)] = med(lambda: f.value_and_grad(P))
            Xc = rng.uniform(0, 1, (args.num_random, D))
            f.maximise_topk(Xc, 8)
            out["topk_ms"] = med(lambda: f.maximise_topk(Xc, 8))
            idx, _ = f.maximise_topk(Xc, 2)
            starts = np.vstack([Xc[np.asarray(idx)], rng.uniform(0, 1, (max(0, args.restarts - 2), D))])
            ctx = sur._context()
            acq, inc, par = f._native_args()
            lo, hi = np.zeros(D), np.ones(D)
            ev = []
            def refine():
                ev.append(ctx.acq_refine(starts, lo, hi, acq, f.scale_factor, inc, par, 200)[3])
            refine()
            out["refine_ms"] = med(refine)
            out["refine_evaluations"] = int(ev[-1])
            def lbfgsb():
                ev.append(ctx.acq_refine(starts, lo, hi, acq, f.scale_factor, inc, par, 15000, lbfgsb=True)[3])
            lbfgsb()
            out["lbfgsb_ms"] = med(lbfgsb)
            out["lbfgsb_evaluations"] = int(ev[-1])
            for mode in ("device", "lbfgsb", "scipy"):
                aux = ta.RandomAndQuasiNewton(num_random=args.num_random, grad_restarts=args.restarts, start_from_best=2,
                                              on_device=(mode == "device"), lockstep="scipy" if mode == "scipy" else True)
                np.random.seed(23)
                aux(b, f)
                res = []
                def run():
                    np.random.seed(23)
                    res.append(aux(b, f))
                out["stage_%s_ms" % mode] = med(run, 5)
                out["max_acq_%s" % mode] = float(res[-1][1]["max_acq"])
            print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
