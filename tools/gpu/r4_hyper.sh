#!/bin/bash
set -e
O=gpurun_out/r4g; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "device_optimizer or one_launch_hyper or hyper_parameter_optimisation or hyper_fit" 2>&1 | tail -15
python - <<'PY'
import json, time, numpy as np, sys
sys.path.insert(0, ".")
import turbo_amd as ta
def med(f, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3
for N, D in ((200, 8), (256, 8), (400, 8), (500, 8), (1000, 8), (2048, 16)):
    rng = np.random.RandomState(N + D)
    X = rng.uniform(0, 1, (N, D)); y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
    ls = float(np.sqrt(D / 6.0))
    out = {"N": N, "D": D}
    for opt in ("fmin_l_bfgs_b", "device"):
        sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, ls, 1e-2), normalize_y=True, optimizer=opt),
                                training_iterations=3, param_continuity=False, incremental=False)
        def fit():
            np.random.seed(11)
            return sur.construct_model(0, X, y)
        fit()
        out[opt + "_ms"] = med(fit, 5)
        m, info = fit()
        out[opt + "_lml"] = float(m.get_log_likelihood()); out[opt + "_evals"] = info["lml_evaluations"]
        sur.close()
    print(json.dumps(out), flush=True)
PY
