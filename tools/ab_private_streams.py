#!/usr/bin/env python3
"""A/B: do other handles' private streams slow down the fits of a handle on the shared stream pair?
(after one side-by-side hyper-parameter fit, fixed-theta fits at N = 1000 / 2048
took 2x in the process that walks N upward.)  Every mode runs in a process of its own -- the ORDER in which
the streams come into being is the variable -- and prints one JSON line per stage:

    python tools/ab_private_streams.py > gpurun_out/ab_private_streams.jsonl

  pair_first          the shared pair (main + background stream) exists before three private streams do
  private_first_kept  three handles on private streams fit first and KEEP their streams (round 3's behaviour
                      after a threaded construct_model at N <= 512, where no fit needs the background stream
                      yet); only then the first fit large enough to create the background stream
  private_first_released   the same, but the private streams are released before that fit (round 4's fix)
  plugin              through HipGPSurrogate: threaded construct_model at N = 500, then fits at N = 1000 / 2048
  two_factories       (round 5) the reference's demos keep several optimisers -- several surrogate factories -- in one
                      process (turbo/modules/surrogates.py:313-324 runs per factory): the hyper-parameter fit of ONE
                      factory at N = 500, 3 starts, both drivers (SciPy threads / optimizer='device'), then the same with
                      a SECOND factory alive that has run its own threaded fits.  Round 4: every factory kept three
                      private streams, the runtime's hardware queues were over-subscribed and the fit took 2x
                      (11.6 -> 19.3 ms); now the device has ONE pool of worker handles (tgp_workers_acquire).
"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
MODES = ("pair_first", "private_first_kept", "private_first_released", "plugin", "two_factories")


def data(N, D=8):
    rng = np.random.RandomState(N + D)
    X = rng.uniform(0, 1, (N, D))
    return X, np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)


def fit_ms(gp, N, reps=15):
    X, y = data(N)
    ls = float(np.sqrt(8 / 6.0))
    ts = []
    for _ in range(reps + 3):
        gp.fit(X, y, "matern52", 1.0, ls, 1e-4, 1e-10, True)
        ts.append(gp.profile_read()["last_fit_ms"])
    return float(np.median(ts[3:]))


def hyper_ms(sur, X, y, reps=7):
    import time
    ts = []
    for r in range(reps + 2):
        np.random.seed(11)
        t0 = time.perf_counter()
        sur.construct_model(r, X, y)
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts[2:]))


def two_factories():
    import turbo_amd as ta
    X, y = data(500)

    def make(opt):
        return ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 1.0, 1e-2), normalize_y=True, optimizer=opt),
                                 training_iterations=3, param_continuity=False, incremental=False)
    for opt in ("fmin_l_bfgs_b", "device"):
        a = make(opt)
        alone = hyper_ms(a, X, y)
        b = make(opt)
        second = hyper_ms(b, X, y)          # the second factory's own fits, the first one's workers still alive
        first_again = hyper_ms(a, X, y)     # ... and the first factory's with the second alive
        c = make("fmin_l_bfgs_b" if opt == "device" else "device")
        hyper_ms(c, X, y, reps=1)           # a third factory on the OTHER driver has run as well
        third_alive = hyper_ms(a, X, y)
        print(json.dumps(dict(mode="two_factories", optimizer=opt, n=500, starts=3, alone_ms=alone, second_factory_ms=second,
                              first_factory_with_second_alive_ms=first_again, first_factory_with_three_alive_ms=third_alive,
                              hw_queues=os.environ.get("GPU_MAX_HW_QUEUES"))), flush=True)
        for f in (a, b, c):
            f.close()


def run(mode):
    import turbo_amd as ta
    if mode == "two_factories":
        return two_factories()
    gp = ta.NativeGP(0, "f64")

    def stage(name, **extra):
        print(json.dumps(dict(mode=mode, stage=name, fit_ms_n1000=fit_ms(gp, 1000), fit_ms_n2048=fit_ms(gp, 2048),
                              hw_queues=os.environ.get("GPU_MAX_HW_QUEUES"), **extra)), flush=True)
    X, y = data(500)

    def workers():
        ws = [ta.NativeGP(0, "f64") for _ in range(3)]
        for w in ws:
            w.set_private_stream(True)
            w.fit(X, y, "matern52", 1.0, 1.0, 1e-4, 1e-10, True)
        return ws
    if mode == "pair_first":
        stage("fresh")
        ws = workers()
        stage("three private streams alive, fitted on")
        for w in ws:
            w.set_private_stream(False)
        stage("private streams released")
    elif mode == "private_first_kept":
        ws = workers()
        stage("first large fits with three private streams alive")
        for w in ws:
            w.set_private_stream(False)
        stage("private streams released afterwards")
    elif mode == "private_first_released":
        ws = workers()
        for w in ws:
            w.set_private_stream(False)
        stage("first large fits after the private streams were released")
    else:
        sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 1.0, 1e-2), normalize_y=True),
                                training_iterations=3, param_continuity=False, incremental=False)
        np.random.seed(11)
        _, info = sur.construct_model(0, X, y)
        stage("after a threaded construct_model (N = 500, 3 starts)", lml_evaluations=info.get("lml_evaluations"))
        sur.close()
        stage("after HipGPSurrogate.close()")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for hwq in (None, "8"):
            for m in MODES:
                env = dict(os.environ)
                if hwq:
                    env["GPU_MAX_HW_QUEUES"] = hwq
                subprocess.run([sys.executable, os.path.abspath(__file__), m], env=env, check=True)
