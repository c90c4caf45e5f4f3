#!/bin/bash
set -e
python -m pytest tests/test_gpu_round4.py -x -q -k "predict_many" 2>&1 | tail -12
python -m pytest tests/test_gpu_parity.py -x -q -k "predict_many or stored_models" 2>&1 | tail -3
python - <<'PY'
import sys, time, numpy as np
sys.path.insert(0, ".")
import turbo_amd as ta
rng = np.random.RandomState(0)
X = rng.uniform(0, 1, (256, 2)); y = np.sin(5 * X[:, 0]) * np.cos(3 * X[:, 1]) + 0.01 * rng.normal(size=256)
grid = rng.uniform(0, 1, (10000, 2))
sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 0.5, 1e-3), optimizer=None, normalize_y=True), training_iterations=1, incremental=False)
for sizes in (list(range(130, 256, 7)), list(range(20, 128, 6))):
    models = [sur.construct_model(t, X[:n], y[:n])[0] for t, n in enumerate(sizes)]
    def many(): return sur.predict_many(models, grid, return_std_dev=True)
    def one(): return [m.predict(grid, return_std_dev=True) for m in models]
    for f in (many, one):
        f(); ts = []
        for _ in range(5):
            t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
        print(len(sizes), "models N=%d..%d" % (sizes[0], sizes[-1]), f.__name__, "%.3f ms" % (1e3 * float(np.median(ts))), "device %.3f" % sur._context().profile_read()["last_sweep_ms"])
PY
