#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Runs only in the build container: it imports mbway/turbo from /root/reference (which never
travels to the GPU box) together with the scikit-learn 1.7.2 / SciPy 1.15.3 it delegates to,
drives the *unmodified* reference classes

    turbo.modules.SciKitGPSurrogate      (turbo/modules/surrogates.py:225-365)
    turbo.modules.EI / PI / UCB          (turbo/modules/acquisition_functions.py:80-358)
    turbo.modules.RandomAndQuasiNewton   (turbo/modules/auxiliary_optimisers.py:16-129)
    turbo.Optimiser                      (turbo/optimiser.py:227-357)

on small seeded inputs and stores inputs + outputs as ``.npz``.  The fixtures are data only
(no reference source).  Re-run with:  python tests/golden/make_golden.py
"""
import os
import sys
import warnings

import numpy as np

REF = "/root/reference"
if not os.path.isdir(os.path.join(REF, "turbo")):
    sys.exit("make_golden.py needs the reference checkout at /root/reference; "
             "the committed .npz fixtures are what travels.")
sys.path.insert(0, REF)

import sklearn  # noqa: E402
import sklearn.gaussian_process as sk_gp  # noqa: E402
import scipy  # noqa: E402

import turbo as tb  # noqa: E402
import turbo.modules as tm  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
K = sk_gp.kernels

# harness-side shim only: numpy >= 1.23 dropped asscalar, which
# turbo/modules/auxiliary_optimisers.py:125 still calls.
if not hasattr(np, "asscalar"):
    np.asscalar = lambda a: np.asarray(a).item()


def synth(seed, N, D):
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 1, size=(N, D))
    w = rng.normal(size=D) / np.sqrt(D)
    y = np.sin(3 * X @ w) + 0.5 * ((X - 0.5) ** 2).sum(1) + 0.01 * rng.normal(size=N)
    return X, y


def candidates(seed, X, M, D):
    """uniform candidates + exact copies of training points + far-away points."""
    rng = np.random.RandomState(seed)
    C = rng.uniform(0, 1, size=(M, D))
    ncopy = min(8, len(X))
    C[:ncopy] = X[:ncopy]                       # sigma -> ~0 / clamp path
    C[ncopy:ncopy + 4] = 50.0 + rng.uniform(0, 1, size=(4, D))   # sigma -> sqrt(c+s2)*y_std
    C[ncopy + 4] = X[0] + 1e-9                  # near-duplicate
    return C


def run_case(name, X, y, kernel, kind, constant, ls, noise, normalize_y=True, M=200,
             jitter=1e-10, cand_seed=7):
    D = X.shape[1]
    sur = tm.SciKitGPSurrogate(
        model_params=dict(kernel=kernel, optimizer=None, normalize_y=normalize_y, alpha=jitter),
        training_iterations=1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model, fitting_info = sur.construct_model(0, X, y)
    gpr = model.model
    Xc = candidates(cand_seed, X, M, D)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mus, sigmas = model.predict(Xc, return_std_dev=True)
        mus_only = model.predict(Xc)
    assert np.array_equal(mus, mus_only)
    out = dict(
        X=X, y=y, kind=kind, constant=constant, length_scale=np.atleast_1d(ls), noise=noise,
        jitter=jitter, normalize_y=normalize_y,
        y_mean=np.float64(np.ravel(gpr._y_train_mean)[0]), y_std=np.float64(np.ravel(gpr._y_train_std)[0]),
        L=gpr.L_, alpha=gpr.alpha_, lml=model.get_log_likelihood(),
        hyper_params=model.get_hyper_params(),
        hyper_param_names=np.array(model.get_hyper_param_names()),
        fitting_iterations=fitting_info["iterations"],
        Xc=Xc, mus=mus, sigmas=sigmas,
    )
    if X.shape[0] <= 16:
        Kfull = gpr.kernel_(X)
        Kfull[np.diag_indices_from(Kfull)] += gpr.alpha
        out["K"] = Kfull
    inc = {"min": float(np.min(y)), "max": float(np.max(y))}
    for ext in ("min", "max"):
        facs = {"ei": tm.EI(xi=0.01), "pi": tm.PI(xi=0.01), "ucb2": tm.UCB(beta=2.0),
                "ucbinf": tm.UCB(beta=float("inf"))}
        for an, fac in facs.items():
            args = [0, model, ext]
            if fac.get_type() == "improvement":
                args.append(inc[ext])
            f, info = fac.construct_function(*args)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                vals = f(Xc)
            out["acq_%s_%s" % (an, ext)] = np.asarray(vals, dtype=np.float64)
            out["name_%s_%s" % (an, ext)] = f.get_name()
        out["incumbent_" + ext] = inc[ext]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "N=%d D=%d M=%d lml=%.6f" % (X.shape[0], D, M, out["lml"]))


def static_cases():
    def iso(D):
        return float(np.sqrt(D / 6.0))

    # RBF iso, 2D / 8D / 32D
    for D, N in ((2, 12), (8, 48), (32, 64)):
        X, y = synth(100 + D, N, D)
        ls = iso(D)
        run_case("rbf_iso_%dd" % D, X, y, K.ConstantKernel(1.0) * K.RBF(ls) + K.WhiteKernel(1e-4),
                 "rbf", 1.0, ls, 1e-4)
    # Matern-5/2 ARD 16D, constant != 1
    D, N = 16, 64
    X, y = synth(216, N, D)
    ls = iso(D) * (0.5 + np.arange(D) / (D - 1.0))
    run_case("matern52_ard_16d", X, y,
             K.ConstantKernel(2.5) * K.Matern(length_scale=ls, nu=2.5) + K.WhiteKernel(1e-4),
             "matern52", 2.5, ls, 1e-4)
    # Matern-3/2 iso 64D, f32-style noise
    D, N = 64, 64
    X, y = synth(364, N, D)
    run_case("matern32_iso_64d", X, y,
             K.ConstantKernel(1.0) * K.Matern(length_scale=iso(D), nu=1.5) + K.WhiteKernel(1e-2),
             "matern32", 1.0, iso(D), 1e-2)
    # Matern-1/2 (exponential) 4D, no WhiteKernel term at all
    D, N = 4, 16
    X, y = synth(404, N, D)
    run_case("matern12_iso_4d_nowhite", X, y,
             K.ConstantKernel(0.7) * K.Matern(length_scale=0.9, nu=0.5),
             "matern12", 0.7, 0.9, 0.0, jitter=1e-6)
    # reference default kernel (turbo/modules/surrogates.py:231-243) on Branin points
    rng = np.random.RandomState(5)
    X = np.hstack([rng.uniform(-5, 10, size=(16, 1)), rng.uniform(0, 15, size=(16, 1))])
    y = branin(X[:, 0], X[:, 1])
    run_case("default_matern52_white_branin", X, y,
             1.0 * K.Matern(nu=2.5) + K.WhiteKernel(),
             "matern52", 1.0, 1.0, 1.0)
    # normalize_y=False and constant y (std -> 1 path, _data.py:92-110)
    D, N = 3, 10
    X, _ = synth(503, N, D)
    run_case("rbf_3d_no_normalise", X, np.linspace(-1, 2, N), K.ConstantKernel(1.3) * K.RBF(0.8) + K.WhiteKernel(1e-3),
             "rbf", 1.3, 0.8, 1e-3, normalize_y=False)
    run_case("rbf_3d_constant_y", X, np.full(N, 4.25), K.ConstantKernel(1.0) * K.RBF(0.8) + K.WhiteKernel(1e-3),
             "rbf", 1.0, 0.8, 1e-3, normalize_y=True)
    # ragged sizes: N not a multiple of any tile, M=1 and M odd
    D, N = 5, 37
    X, y = synth(605, N, D)
    run_case("rbf_5d_ragged", X, y, K.ConstantKernel(1.0) * K.RBF(iso(D)) + K.WhiteKernel(1e-4),
             "rbf", 1.0, iso(D), 1e-4, M=77)
    D, N = 2, 1
    X, y = synth(701, N, D)
    run_case("rbf_2d_single_point", X, y, K.ConstantKernel(1.0) * K.RBF(0.5) + K.WhiteKernel(1e-4),
             "rbf", 1.0, 0.5, 1e-4, M=15)


def branin(x, y):
    # demos/Branin-Hoo.ipynb cell 5
    from math import pi
    return (y - (5.1 / (4 * pi ** 2)) * x ** 2 + 5 * x / pi - 6) ** 2 + 10 * (1 - 1 / (8 * pi)) * np.cos(x) + 10


def not_pd_case():
    """duplicate rows + zero noise + zero jitter -> LinAlgError (sklearn _gpr.py:350-358)."""
    X, y = synth(800, 6, 2)
    X[3] = X[1]
    sur = tm.SciKitGPSurrogate(
        model_params=dict(kernel=K.ConstantKernel(1.0) * K.RBF(0.5), optimizer=None,
                          normalize_y=True, alpha=0.0), training_iterations=1)
    try:
        sur.construct_model(0, X, y)
        raised = ""
    except np.linalg.LinAlgError as e:
        raised = type(e).__name__
    assert raised == "LinAlgError"
    np.savez_compressed(os.path.join(HERE, "not_pd.npz"), X=X, y=y, kind="rbf", constant=1.0,
                        length_scale=np.array([0.5]), noise=0.0, jitter=0.0, raised=raised)
    print("wrote not_pd")


def branin_trace():
    """Config 0: an end-to-end ``Optimiser.run`` on Branin-Hoo through the reference loop.

    seed 42 (demos/Branin-Hoo.ipynb cell 4), LHS pre-phase of 4, fixed default-kernel
    hyper-parameters, EI xi=0.01, RandomAndQuasiNewton(num_random=1024, grad_restarts=0,
    start_from_best=0) -- the gradient stage cannot run under SciPy 1.15.
    Records, per Bayes trial, the training set, the candidate batch the reference drew from the
    global NumPy RNG, and the (x, max_acq) it selected.
    """
    np.random.seed(42)
    bounds = [('x', -5., 10.), ('y', 0., 15.)]
    op = tb.Optimiser(lambda x, y: float(branin(x, y)), 'min', bounds, pre_phase_trials=4,
                      settings_preset=None)
    op.latent_space = tm.NoLatentSpace()
    op.pre_phase_select = tm.LHS_selector(num_total=4)
    op.fallback = tm.Fallback(selector=tm.random_selector())
    op.aux_optimiser = tm.RandomAndQuasiNewton(num_random=1024, grad_restarts=0, start_from_best=0)
    op.surrogate = tm.SciKitGPSurrogate(model_params=dict(
        kernel=1.0 * K.Matern(nu=2.5) + K.WhiteKernel(), normalize_y=True, optimizer=None),
        training_iterations=1)
    op.acquisition = tm.EI(xi=0.01)

    drawn = []
    gen = op.aux_optimiser.gen_random

    def recording_gen(num_points, latent_bounds):
        c = gen(num_points, latent_bounds)
        drawn.append(c.copy())
        return c
    op.aux_optimiser.gen_random = recording_gen

    class Rec(tm.Listener):
        def __init__(self):
            self.sel = {}

        def selection_finished(self, trial_num, x, selection_info):
            self.sel[trial_num] = (np.array(x, copy=True), dict(selection_info))
    rec = Rec()
    op.register_listener(rec)
    max_trials = 33      # last Bayes trial is fitted on N=32 observations
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        op.run(max_trials=max_trials)

    xs = np.vstack(op.rt.trial_xs)
    ys = np.array(op.rt.trial_ys)
    out = dict(trial_xs=xs, trial_ys=ys, bounds_lo=np.array([-5., 0.]), bounds_hi=np.array([10., 15.]),
               kind="matern52", constant=1.0, length_scale=np.array([1.0]), noise=1.0, jitter=1e-10,
               xi=0.01, num_random=1024, pre_phase=4)
    trials, sel_x, max_acq, types, lml = [], [], [], [], []
    k = 0
    for t in range(4, max_trials):
        x, info = rec.sel[t]
        trials.append(t)
        types.append(info['type'])
        # 'bayes_x' holds the Bayes choice when the too-close fallback fired
        bx = info.get('bayes_x', x)
        sel_x.append(np.asarray(bx).reshape(-1))
        max_acq.append(info['maximisation_info']['max_acq'])
        lml.append(info['model'].get_log_likelihood())
        out["cand_%d" % t] = drawn[k]
        k += 1
    assert k == len(drawn)
    out.update(trials=np.array(trials), sel_x=np.vstack(sel_x), max_acq=np.array(max_acq),
               types=np.array(types), lml=np.array(lml))
    np.savez_compressed(os.path.join(HERE, "branin_trace.npz"), **out)
    print("wrote branin_trace: %d bayes trials, best y %.5f" % (len(trials), ys.min()))


if __name__ == "__main__":
    print("reference:", REF, "| sklearn", sklearn.__version__, "| scipy", scipy.__version__,
          "| numpy", np.__version__)
    static_cases()
    not_pd_case()
    branin_trace()
