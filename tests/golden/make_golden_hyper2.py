#!/usr/bin/env python3
"""More optimisation traces FROM THE REFERENCE (round 5: the hyper-parameter fit now runs L-BFGS-B inside the library,
tgp_fit_lbfgsb), generated in the build container like make_golden_hyper.py: the unmodified ``SciKitGPSurrogate``
(turbo/modules/surrogates.py:294-326) with ``training_iterations > 0`` and ``param_continuity``:
  opt_matern32_iso_5d_mid    sizes either side of the one-launch limit (N = 100, 150, 220), three starts
  opt_fixed_noise_3d         WhiteKernel(noise_level_bounds="fixed"): the fixed entry is not in theta
  opt_fixed_constant_ard_3d  ConstantKernel(constant_value_bounds="fixed") * RBF(ARD) + WhiteKernel
  opt_nowhite_matern52_3d    no WhiteKernel at all (alpha = 1e-3 on the diagonal instead): theta has two entries
Data only."""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_hyper import K, opt_case, synth  # noqa: E402  (refuses to run without /root/reference)

if __name__ == "__main__":
    X, y = synth(31, 220, 5)
    opt_case("opt_matern32_iso_5d_mid", X, y, K.ConstantKernel(1.0) * K.Matern(0.9, nu=1.5) + K.WhiteKernel(1e-2), "matern32", 3, [100, 150, 220])
    X, y = synth(32, 140, 3)
    opt_case("opt_fixed_noise_3d", X, y, K.ConstantKernel(1.0) * K.Matern(0.8, nu=2.5) + K.WhiteKernel(1e-2, noise_level_bounds="fixed"),
             "matern52", 3, [40, 140])
    X, y = synth(33, 160, 3)
    opt_case("opt_fixed_constant_ard_3d", X, y, K.ConstantKernel(1.0, constant_value_bounds="fixed") * K.RBF(np.ones(3)) + K.WhiteKernel(1e-2),
             "rbf", 2, [60, 160])
    # a kernel without a noise term: opt_case with model_params['alpha'] (the jitter carries the noise)
    import turbo.modules as tm
    X, y = synth(34, 150, 3)
    name, alpha, iters, sizes = "opt_nowhite_matern52_3d", 1e-3, 3, [50, 150]
    sur = tm.SciKitGPSurrogate(model_params=dict(kernel=K.ConstantKernel(1.0) * K.Matern(0.8, nu=2.5), normalize_y=True, random_state=0,
                                                 alpha=alpha), training_iterations=iters, param_continuity=True)
    out = dict(X=X, y=y, kind="matern52", iters=iters, sizes=np.array(sizes), alpha=alpha)
    for t, n in enumerate(sizes):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model, info = sur.construct_model(t, X[:n], y[:n])
        out["hp_%d" % t] = model.get_hyper_params()
        out["lml_%d" % t] = model.get_log_likelihood()
        out["theta_%d" % t] = model.model.kernel_.theta
        out["bounds"] = model.model.kernel_.bounds
        mu, sg = model.predict(X[-16:], return_std_dev=True)
        out["mu_%d" % t] = mu
        out["sigma_%d" % t] = sg
        print(name, "trial", t, "n", n, "hp", out["hp_%d" % t], "lml %.6f" % out["lml_%d" % t])
    out["names"] = np.array(model.get_hyper_param_names())
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
