import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import turbo_amd as ta
for N in [int(a) for a in sys.argv[1:]]:
    rng = np.random.RandomState(47)
    X = rng.uniform(0, 1, (N, 5)); y = np.sin(4 * X[:, 0]) + X[:, 1] * X[:, 2] + 0.05 * rng.normal(size=N)
    gp = ta.NativeGP(0, "f64")
    t0 = time.time()
    try:
        lml, ym, ys = gp.fit(X, y, "rbf", 1.0, 0.3, 1e-2, 1e-10, True)
        print(N, "ok lml", lml, "%.2fs wall, fit %.2f ms on the GPU" % (time.time() - t0, gp.profile_read()["last_fit_ms"]), flush=True)
    except Exception as e:
        print(N, "FAILED", str(e)[-90:], flush=True)
    del gp
