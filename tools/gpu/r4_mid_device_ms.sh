#!/bin/bash
set -e
python - <<'PY'
import sys, time, os, subprocess, json
import numpy as np
sys.path.insert(0, ".")
code = r'''
import sys, time, numpy as np
sys.path.insert(0, ".")
import turbo_amd as ta
for N in (200, 256, 300, 400, 500):
    rng = np.random.RandomState(N); X = rng.uniform(0, 1, (N, 8)); y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
    gp = ta.NativeGP(0, "f64"); gp.fit(X, y, "matern52", 1.0, 1.1, 1e-4, 1e-10, True)
    out = []
    for M in (1000, 4096, 10000, 16384, 32768):
        Xc = rng.uniform(0, 1, (M, 8)); gp.set_candidates(Xc)
        ts = []
        for _ in range(15):
            r = gp.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01); ts.append(r["sweep_ms"])
        out.append("%d:%.3f" % (M, float(np.median(ts[3:]))))
    print("N=%d" % N, " ".join(out), flush=True)
'''
for env in ({}, {"TGP_MID": "0"}, {"TGP_MID_MAXM": "100000000"}):
    print(env, flush=True)
    subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), check=True)
PY
