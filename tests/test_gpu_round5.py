"""Round 5 (GPU): the sweep that starts inside the fit (tgp_set_overlap), the posterior mean inside the contraction,
and the size ladder of the fit against the oracle.

Reference path: turbo/optimiser.py:336-340 (construct_model, then the acquisition maximised over one vectorised
batch, turbo/modules/auxiliary_optimisers.py:59-66); the arithmetic is sklearn's _gpr.py:443-494 behind
turbo/modules/surrogates.py:332-338."""
import hashlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _synth(seed, N, D, M):
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 1, (N, D))
    y = np.sin(3 * X.sum(1)) + 0.3 * ((X - 0.4) ** 2).sum(1) + 0.01 * rng.normal(size=N)
    Xc = rng.uniform(0, 1, (M, D))
    return X, y, Xc


def _digest(r):
    return hashlib.sha256(r["mu"].tobytes() + r["sigma"].tobytes() + r["acq"].tobytes()).hexdigest(), r["best_idx"], r["best_val"]


# (dtype, kind, N, D, M): 128-row tiles for the whole sweep | 256 x 128 f64 tiles | 256 x 256 f32 tiles | a ragged N whose
# last 256-row unit is half empty | two launch pairs (the second one never starts early)
OVERLAP_CASES = [
    ("f64", "matern52", 700, 6, 5000, False),
    ("f64", "rbf", 2048, 8, 9000, False),
    ("f32", "rbf", 2048, 16, 20000, False),
    ("f32", "matern32", 1930, 5, 17000, True),
    ("f64", "matern12", 1100, 3, 110000, False),
    ("f32", "rbf", 4096, 32, 32768, False),
    ("f64", "matern52", 2048, 16, 70000, True),      # C2's shape (ARD length scales, two launch pairs)
]


@pytest.mark.parametrize("dtype,kind,N,D,M,ard", OVERLAP_CASES)
def test_a_sweep_started_inside_the_fit_is_the_serial_sweep_bit_for_bit(dtype, kind, N, D, M, ard):
    """tgp_set_overlap modes 1 and 2 against mode 0 on the same handle: every mean, deviation and acquisition value, the
    winner and its value are the same BYTES (same kernels, same order of every sum -- only the schedule differs), and
    they are the oracle's to the tolerance of the dtype.  Repeated, so that a front left over from the previous
    step meets the next fit."""
    import turbo_amd as ta
    from oracle import gp_oracle as o
    X, y, Xc = _synth(11 + N + D, N, D, M)
    ls = np.sqrt(D / 6.0) * ((0.5 + np.arange(D) / (D - 1.0)) if ard else 1.0)
    noise = 1e-2 if dtype == "f32" else 1e-4
    inc = float(y.min())
    gp = ta.NativeGP(0, dtype)
    gp.fit(X, y, kind, 1.2, ls, noise, 1e-10, True)
    gp.set_candidates(Xc)
    got = {}
    for mode in (0, 1, 2, 0, 2):
        gp.set_overlap(mode)
        for rep in range(2):
            lml = gp.fit(X, y, kind, 1.2, ls, noise, 1e-10, True)[0]
            r = gp.sweep(ta._lib.ACQ_EI, -1.0, inc, 0.01, want_mu=True, want_sigma=True, want_acq=True)
            d = (_digest(r), lml)
            assert got.setdefault("ref", d) == d, (mode, rep)
    assert r["best_idx"] == int(np.argmax(r["acq"]))
    om = o.fit(X, y, kind, 1.2, ls, noise, 1e-10, True)
    sub = np.random.RandomState(1).choice(M, size=min(M, 3000), replace=False)
    mu, sd = o.predict(om, Xc[sub])
    if dtype == "f64":
        np.testing.assert_allclose(r["mu"][sub], mu, rtol=1e-5, atol=1e-7 * om.y_std)
        np.testing.assert_allclose(r["sigma"][sub] ** 2, sd ** 2, rtol=1e-5, atol=1e-9 * (1.2 + noise) * om.y_std ** 2)
    else:   # f32 sweep: the tolerances of DESIGN.md section 2
        assert np.max(np.abs(r["mu"][sub] - mu)) < 5e-4 * om.y_std
        assert np.max(np.abs(r["sigma"][sub] ** 2 - sd ** 2)) < 5e-5 * (1.2 + noise) * om.y_std ** 2


def test_a_front_nobody_comes_for_is_discarded():
    """Whatever happens between the fit and the sweep -- new candidates (same count, same buffer), a top-k sweep, a
    predict of other points, a second fit with other hyper-parameters, an appended observation -- the sweep that
    follows returns what a strictly serial handle returns."""
    import turbo_amd as ta
    N, D, M = 1500, 7, 12000
    X, y, Xc = _synth(5, N, D, M)
    Xc2 = _synth(6, N, D, M)[2]
    inc = float(y.min())
    a, b = ta.NativeGP(0, "f64"), ta.NativeGP(0, "f64")
    a.set_overlap(2)
    b.set_overlap(0)

    def both(fn):
        ra, rb = fn(a), fn(b)
        return ra, rb

    def sweep(g):
        return _digest(g.sweep(ta._lib.ACQ_EI, -1.0, inc, 0.01, want_mu=True, want_sigma=True, want_acq=True))

    for g in (a, b):
        g.fit(X, y, "rbf", 1.0, 0.9, 1e-3, 1e-10, True)
        g.set_candidates(Xc)
    # 1. new candidates between fit and sweep
    for g in (a, b):
        g.fit(X, y, "rbf", 1.0, 0.9, 1e-3, 1e-10, True)
        g.set_candidates(Xc2)
    ra, rb = both(sweep)
    assert ra == rb
    # 2. a top-k sweep and a predict of other points in between
    for g in (a, b):
        g.fit(X, y, "rbf", 1.0, 0.8, 1e-3, 1e-10, True)
        g.sweep_topk(8, ta._lib.ACQ_UCB, -1.0, 0.0, 2.0)
        g.evaluate(Xc[:300], ta._lib.ACQ_NONE, 1.0, 0.0, 0.0, True, True, False)
    ra, rb = both(sweep)
    assert ra == rb
    # 3. two fits in a row (the first one's front belongs to a factor that is gone), then two sweeps in a row (the
    #    second finds no front)
    for g in (a, b):
        g.fit(X, y, "rbf", 1.0, 0.8, 1e-3, 1e-10, True)
        g.fit(X, y, "matern52", 1.3, 1.1, 1e-3, 1e-10, True)
    assert both(sweep)[0] == both(sweep)[1]
    ra, rb = both(sweep)
    assert ra == rb
    # 4. an appended observation
    Xn = np.vstack([X, Xc[:1]])
    yn = np.append(y, 0.3)
    for g in (a, b):
        g.fit(Xn, yn, "matern52", 1.3, 1.1, 1e-3, 1e-10, True, append=True)
        assert g.appended
    ra, rb = both(sweep)
    assert ra == rb
    # 5. a kernel matrix that is not positive definite leaves no usable front behind
    Xbad = np.vstack([X[:1499], X[:1]])
    for g in (a, b):
        with pytest.raises(np.linalg.LinAlgError):
            g.fit(Xbad, y, "rbf", 1.0, 0.9, 0.0, 0.0, True)
        g.fit(X, y, "rbf", 1.0, 0.9, 1e-3, 1e-10, True)
    ra, rb = both(sweep)
    assert ra == rb
    # 6. two handles of the device share its third stream: fronts issued back to back, swept in the other order
    a2 = ta.NativeGP(0, "f64")
    a2.set_overlap(2)
    a2.fit(X, y, "matern52", 1.3, 1.1, 1e-3, 1e-10, True)
    a2.set_candidates(Xc2)
    for g in (a, a2, b):
        g.fit(X, y, "rbf", 1.0, 0.9, 1e-3, 1e-10, True)
    r2 = _digest(a2.sweep(ta._lib.ACQ_EI, -1.0, inc, 0.01, want_mu=True, want_sigma=True, want_acq=True))
    ra, rb = both(sweep)
    assert ra == rb
    b.set_candidates(Xc2)
    assert r2 == sweep(b)
    a2.close()
    a.close()   # (with a front possibly still in flight: destroy waits for it)
    b.close()


def test_a_candidates_mean_does_not_depend_on_the_tile_variant_or_the_batch_it_travels_in():
    """The posterior mean is now accumulated inside the contraction (MeanAcc, csrc/mfma_gemm.hpp): four partial sums per
    candidate in ONE order for every tile variant.  The same 3000 candidates as a batch of their own (128 x 128 tiles),
    at the head of a batch of 40000 (256-row tiles) and as a shard of it: the same bytes."""
    import turbo_amd as ta
    for dtype in ("f64", "f32"):
        N, D = 2300, 9
        X, y, Xc = _synth(17, N, D, 40000)
        gp = ta.NativeGP(0, dtype)
        gp.fit(X, y, "matern52", 1.0, 1.2, 1e-3, 1e-10, True)
        out = []
        for lo, hi in ((0, 3000), (0, 40000), (0, 20000)):
            gp.set_candidates(Xc[lo:hi])
            r = gp.sweep(ta._lib.ACQ_UCB, -1.0, 0.0, 2.0, want_mu=True, want_sigma=True, want_acq=True)
            out.append((r["mu"][:3000].tobytes(), r["sigma"][:3000].tobytes(), r["acq"][:3000].tobytes()))
        assert out[0] == out[1] == out[2], dtype


def test_tuning_table_and_overlap_argument_checks():
    import turbo_amd as ta
    t = ta._lib.tuning()
    assert t["TGP_OVERLAP"][0] in ("0", "1", "2") and "TGP_PANEL_FUSE_TILES" in t and t["TGP_SLAB_GB"][0] == "1"
    gp = ta.NativeGP(0, "f64")
    with pytest.raises(ValueError):
        gp.set_overlap(3)
    gp.set_overlap(2)     # no candidates, no fit: harmless
    X, y, Xc = _synth(3, 300, 4, 100)
    gp.fit(X, y, "rbf", 1.0, 1.0, 1e-3, 1e-10, True)     # no candidates resident: nothing starts early
    gp.set_candidates(Xc)
    r = gp.sweep(ta._lib.ACQ_NONE, want_mu=True)
    assert np.all(np.isfinite(r["mu"]))


def test_predict_many_serves_a_foreign_model_beside_native_ones():
    """round-4 advisor: predict_many looked at m.X of EVERY model; the reference's Surrogate.ModelInstance only carries
    .model (turbo/modules/surrogates.py:328-338), so a list that mixes this package's models with foreign ones raised
    AttributeError.  Foreign models are served one by one through their own predict."""
    import turbo_amd as ta

    class Foreign:
        def predict(self, X, return_std_dev=False):
            mu = X[:, 0] * 2.0
            return (mu, np.full(len(X), 0.5)) if return_std_dev else mu

    X, y, Xq = _synth(9, 60, 3, 50)
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("rbf", 1.0, 0.8, 1e-4), optimizer=None, normalize_y=True),
                            training_iterations=1)
    m1, _ = sur.construct_model(0, X[:40], y[:40])
    m2, _ = sur.construct_model(1, X, y)
    models = [m1, Foreign(), m2]
    mus, sig = sur.predict_many(models, Xq, return_std_dev=True)
    for t, m in enumerate(models):
        mu, sd = m.predict(Xq, return_std_dev=True)
        np.testing.assert_allclose(mus[t], np.asarray(mu).reshape(-1), rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(sig[t], np.asarray(sd).reshape(-1), rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(sur.predict_many(models, Xq)[1], Xq[:, 0] * 2.0)


# ---- the size ladder of the fit (round 4's tools/gpu/r4_ladder.sh as a parity test) ------------------------------
# One N inside EVERY padded size class Np = 256 ... 12288 (Np = ceil(N / 256) * 256 decides the outer block, the panel
# launches, which blocks are fused, the inverse's schedule): LML and alpha against the oracle (LAPACK dpotrf / cho_solve
# behind sklearn _gpr.py:349-364, reached from turbo/modules/surrogates.py:318).  Named cases: Np = 6912, 7936, 8960 (outer
# blocks of 1024 would leave a last block of 768 -- `invalid configuration argument` in round 4) and Np > 9216 (the
# inverse level by level instead of behind the panel chain).  One fit per class, no repetition.
_LADDER = [(np_, np_ - (37 if (np_ // 256) % 3 else (0 if (np_ // 256) % 2 else 255))) for np_ in range(256, 12289, 256)]
_LADDER_GP = {}


@pytest.mark.parametrize("Np,N", _LADDER, ids=["Np%d-N%d" % c for c in _LADDER])
def test_fit_ladder_against_the_oracle(Np, N):
    import turbo_amd as ta
    from oracle import gp_oracle as o
    assert (N + 255) // 256 * 256 == Np
    rng = np.random.RandomState(N)
    X = rng.uniform(0, 1, (N, 6))
    y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
    gp = _LADDER_GP.setdefault("gp", ta.NativeGP(0, "f64"))     # one handle, sizes ascending: buffers only grow
    lml = gp.fit(X, y, "matern52", 1.1, 0.8, 1e-3, 1e-10, True)[0]
    om = o.fit(X, y, "matern52", 1.1, 0.8, 1e-3, 1e-10, True)
    assert abs(lml - om.lml) <= 1e-9 * abs(om.lml), (lml, om.lml)
    np.testing.assert_allclose(gp.debug_read(ta._lib.BUF_ALPHA), om.alpha, rtol=1e-6, atol=1e-7 * np.abs(om.alpha).max())
    if Np == 12288:
        _LADDER_GP.pop("gp").close()


def test_a_second_factory_with_live_workers_does_not_slow_the_hyper_parameter_fit():
    """more streams alive in the process than hardware queues -> two share a queue and run one
    after the other (two factories with three private worker streams each: N = 500, 11.6 -> 19.3 ms).  Now one pool of
    worker handles per device (tgp_workers_acquire) and GPU_MAX_HW_QUEUES=8 by default: the hyper-parameter fit of a
    factory (turbo/modules/surrogates.py:313-318, 3 starts) takes the same time with a second and a third factory alive,
    for both drivers.  Runs in a process of its own (tools/ab_private_streams.py two_factories)."""
    import json
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ab_private_streams.py"), "two_factories"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    recs = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(recs) == 2
    for r in recs:
        for key in ("second_factory_ms", "first_factory_with_second_alive_ms", "first_factory_with_three_alive_ms"):
            assert r[key] <= 1.3 * r["alone_ms"], (r["optimizer"], key, r[key], r["alone_ms"])


def test_the_worker_pool_is_one_per_device_and_refuses_misuse():
    import turbo_amd as ta
    a, b = ta.NativeGP(0, "f64"), ta.NativeGP(0, "f32")
    with a.workers(3) as wa:
        ha = [w._h.value for w in wa]
        X, y, _ = _synth(2, 300, 4, 1)
        assert np.isfinite(wa[1].fit(X, y, "rbf", 1.0, 1.0, 1e-3, 1e-10, True)[0])
        assert wa[0].lib.tgp_destroy(wa[0]._h) == ta._lib.BAD_ARG          # a worker belongs to the library
    with b.workers(4) as wb:
        assert [w._h.value for w in wb][:3] == ha                          # the same pool whoever asks
    with pytest.raises(ValueError):
        with a.workers(5):
            pass


def test_candidate_sweep_prefetch_next_chooses_the_same_points():
    """The overlap through the reference's plugin API: CandidateSweep(device_rng_seed=..., prefetch_next=True) draws the next
    trial's batch behind this trial's sweep, HipGPSurrogate's next fit starts that batch's sweep inside itself
    (turbo/optimiser.py:336-340: construct_model, construct_function, aux_optimiser back to back).  A short optimisation
    loop over growing data: the chosen points and their values are those of prefetch_next=False, bit for bit."""
    import turbo_amd as ta
    b = ta.Bounds([("x%d" % d, 0.0, 1.0) for d in range(5)])
    rng = np.random.RandomState(4)
    X0 = rng.uniform(0, 1, (700, 5))

    def f(x):
        return np.sin(3 * x.sum(-1)) + 0.3 * ((x - 0.4) ** 2).sum(-1)

    chosen = {}
    for prefetch in (False, True):
        sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 0.9, 1e-3), optimizer=None, normalize_y=True),
                                training_iterations=1, incremental=False)
        aux = ta.CandidateSweep(num_random=30000, device_rng_seed=123, prefetch_next=prefetch)
        X, y = X0.copy(), f(X0)
        pts = []
        for trial in range(5):
            model, _ = sur.construct_model(trial, X, y)
            acq, _ = ta.EI(0.01).construct_function(trial, model, "min", float(y.min()))
            x, info = aux(b, acq)
            pts.append((x.tobytes(), info["max_acq"]))
            if trial == 2:
                model.predict(X[:10])            # something else uses the context in between: the prefetched batch is replaced
            X, y = np.vstack([X, x]), np.append(y, f(x))
        chosen[prefetch] = pts
        sur.close()
    assert chosen[False] == chosen[True]


@pytest.mark.parametrize("kind,N,D,ard,iters", [("matern52", 12, 2, False, 3), ("rbf", 64, 4, True, 4), ("matern32", 100, 3, False, 3),
                                                ("matern52", 128, 8, True, 2), ("matern12", 129, 2, False, 3), ("rbf", 700, 6, True, 3),
                                                ("matern52", 1700, 4, False, 2)])
def test_the_default_hyper_parameter_fit_walks_what_scipy_walks(kind, N, D, ard, iters):
    """optimizer='fmin_l_bfgs_b' (the default) is L-BFGS-B inside the library (tgp_fit_lbfgsb, csrc/host_lbfgsb.hpp): at
    every size -- one-launch evaluations at N <= 128, a chain of launches above, one thread above N = 1536 -- the same
    optimum, after as many objective evaluations (give or take the cases where rounding decides a trial step), as SciPy's
    L-BFGS-B driving the same GPU objective from Python (optimizer='scipy'), which is what GaussianProcessRegressor.fit does (_gpr.py:296-337, :654-670)."""
    import warnings
    import turbo_amd as ta
    rng = np.random.RandomState(N + D)
    X = rng.uniform(0, 1, (N, D))
    y = np.sin(3 * X @ rng.normal(size=D)) + 0.5 * ((X - 0.5) ** 2).sum(1) + 0.05 * rng.normal(size=N)
    got = {}
    for opt in ("scipy", "fmin_l_bfgs_b"):
        k = ta.GPKernel(kind, 1.0, np.ones(D) if ard else 1.0, 1e-2)
        sur = ta.HipGPSurrogate(model_params=dict(kernel=k, normalize_y=True, random_state=0, optimizer=opt),
                                training_iterations=iters, param_continuity=False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model, info = sur.construct_model(0, X, y)
        got[opt] = (model.get_log_likelihood(), np.log(model.get_hyper_params()), info["lml_evaluations"])
        sur.close()
    (l_ref, t_ref, e_ref), (l_lib, t_lib, e_lib) = got["scipy"], got["fmin_l_bfgs_b"]
    # (equal counts in six of these seven cases when this was written; a last-bit difference in the optimisers' own
    # arithmetic can grow into a different trial step on this objective -- see test_device_optimizer_above_128)
    assert 0.6 * e_ref - 10 <= e_lib <= 1.5 * e_ref + 25, (e_lib, e_ref)
    assert abs(l_lib - l_ref) <= 1e-8 * max(1.0, abs(l_ref)), (l_lib, l_ref)
    np.testing.assert_allclose(t_lib, t_ref, atol=5e-3)     # (flat optima: the LML above is the sharp check)


def test_callables_go_through_scipy_and_fixed_hyper_parameters_through_the_library():
    """a callable optimizer is handed the objective as scikit-learn hands it over (SciPy drives the GPU).  A kernel with
    a FIXED hyper-parameter has fewer than 2 + n_ls entries in theta: the library takes the full vector with lo == hi at
    the fixed entry and optimises the others -- what SciPy walks on the reduced theta (optimizer='scipy'), not a box
    coordinate that cannot move."""
    import warnings
    import turbo_amd as ta
    X, y, _ = _synth(3, 150, 3, 1)
    calls = []

    def my_opt(obj_func, initial_theta, bounds):
        import scipy.optimize
        calls.append(1)
        r = scipy.optimize.minimize(obj_func, initial_theta, method="L-BFGS-B", jac=True, bounds=bounds)
        return r.x, r.fun
    out = []
    for opt in ("scipy", my_opt):
        sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 0.8, 1e-2), normalize_y=True,
                                                  random_state=0, optimizer=opt), training_iterations=2)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model, info = sur.construct_model(0, X, y)
        out.append((model.get_log_likelihood(), info["lml_evaluations"]))
        sur.close()
    assert len(calls) == 2 and out[0] == out[1]
    for fixed, ard in (("noise", False), ("constant", True), ("length_scale", False)):
        res = []
        for opt in ("fmin_l_bfgs_b", "scipy", "device"):
            k = ta.GPKernel("matern52", 1.3, np.full(3, 0.8) if ard else 0.8, 2e-2, bounds={fixed: "fixed"})
            sur = ta.HipGPSurrogate(model_params=dict(kernel=k, normalize_y=True, random_state=0, optimizer=opt), training_iterations=3)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                model, info = sur.construct_model(0, X, y)
            hp = dict(zip(model.get_hyper_param_names(), model.get_hyper_params()))
            res.append((model.get_log_likelihood(), info["lml_evaluations"], model.get_hyper_params(), len(hp)))
            sur.close()
        (l_lib, e_lib, t_lib, n_lib), (l_ref, e_ref, t_ref, n_ref), (l_dev, e_dev, t_dev, n_dev) = res
        assert n_lib == n_ref == (4 if ard else 2), (fixed, n_lib)          # the fixed one is not among the hyper-parameters
        assert abs(l_lib - l_ref) <= 1e-8 * max(1.0, abs(l_ref)), (fixed, l_lib, l_ref)
        np.testing.assert_allclose(np.log(t_lib), np.log(t_ref), atol=5e-3)
        assert 0.6 * e_ref - 10 <= e_lib <= 1.5 * e_ref + 25, (fixed, e_lib, e_ref)
        assert (l_dev, e_dev) == (l_lib, e_lib)                               # N > 128: 'device' is the same path


@pytest.mark.parametrize("N", [1100, 2300, 3000])      # Np = 1280 (outer blocks of 256), 2304 (512 + a tail of 256), 3072 (512)
def test_a_fit_on_a_private_stream_is_the_same_with_or_without_the_background_stream_on_loan(N):
    """a handle on a private stream (the workers of a hyper-parameter fit) runs its share of the inverse in line after the
    last panel -- every outer block's own inverse in batched launches, then block by block what depends on the blocks
    before -- unless it is the only such fit in flight, when it borrows the device's background stream and issues what a
    handle on the shared stream issues (csrc/fit_kernels.hip private_fit_begin, inverse_inner_all).  The same
    operations either way: alone (on loan), three side by side (in line, or on loan when the others happen to be
    between two fits), with the loan switched off -- the same bytes every time, and the shared stream's."""
    import subprocess
    import sys
    import threading
    import turbo_amd as ta
    X, y, Xc = _synth(77, N, 5, 3000)

    def run(g):
        lml = g.fit(X, y, "matern52", 1.2, 0.7, 1e-3, 1e-10, True)[0]
        g.set_candidates(Xc)
        r = g.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)
        lg, grad = g.fit_grad(X, y, "matern52", 1.2, 0.7, 1e-3, 1e-10, True)
        return _digest(r), lml, lg, grad.tobytes()
    shared = ta.NativeGP(0, "f64")
    ref = run(shared)
    with shared.workers(3) as ws:
        assert run(ws[0]) == ref                  # alone: on loan
        out = [[] for _ in ws]

        def work(i):
            for _ in range(4):
                out[i].append(run(ws[i]))
        ths = [threading.Thread(target=work, args=(i,)) for i in range(len(ws))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert all(o == ref for per in out for o in per)
    child = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
             "import turbo_amd as ta\nfrom test_gpu_round5 import _synth, _digest\n"
             "X, y, Xc = _synth(77, %d, 5, 3000)\ng = ta.NativeGP(0, 'f64')\n"
             "assert ta._lib.tuning()['TGP_BG_LEASE'][0] == '0'\n"
             "with g.workers(1) as ws:\n"
             "    w = ws[0]\n"
             "    lml = w.fit(X, y, 'matern52', 1.2, 0.7, 1e-3, 1e-10, True)[0]\n"
             "    w.set_candidates(Xc)\n"
             "    r = w.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)\n"
             "    print(_digest(r)[0], repr(lml))\n" % (ROOT, os.path.join(ROOT, "tests"), N))
    res = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, TGP_BG_LEASE="0"), capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    assert res.stdout.split() == [ref[0][0], repr(ref[1])]


@pytest.mark.parametrize("kind,N,D,acq_name", [("matern52", 30, 2, "ei"), ("rbf", 100, 4, "ucb"), ("matern32", 300, 3, "pi"),
                                               ("matern52", 900, 6, "ei"), ("rbf", 1500, 8, "ucb")])
def test_the_gradient_stage_in_the_library_walks_what_scipy_walks(kind, N, D, acq_name):
    """tgp_acq_lbfgsb -- L-BFGS-B per restart inside the library, the restarts in lock-step over one batched closed-form
    gradient evaluation per round -- against SciPy's L-BFGS-B driving tgp_acq_grad one restart at a time (the reference's
    gradient stage, turbo/modules/auxiliary_optimisers.py:80-99, with the closed-form gradient): the same end point and
    value per restart, as many evaluations (give or take a restart where rounding decides a trial step), the same
    success flags; and through CandidateSweep the same chosen point as the Python lock-step of rounds 2-4."""
    import scipy.optimize
    import turbo_amd as ta
    X, y, _ = _synth(9 + N, N, D, 1)
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel(kind, 1.2, 0.6, 1e-3), optimizer=None, normalize_y=True),
                            training_iterations=1)
    model, _ = sur.construct_model(0, X, y)
    fac = {"ei": ta.EI(0.01), "pi": ta.PI(0.01), "ucb": ta.UCB(2.0)}[acq_name]
    args = () if acq_name == "ucb" else (float(y.min()),)
    f, _ = fac.construct_function(0, model, "min", *args)
    bounds = [(0.0, 1.0)] * D
    R = 8
    starts = np.random.RandomState(4).uniform(0, 1, (R, D))
    xs, vs, st, evals = f.lbfgsb(starts, bounds)
    ev_ref, n_same = 0, 0
    for j in range(R):
        count = [0]

        def neg_f(x):
            count[0] += 1
            v, g = f.value_and_grad(x.reshape(1, -1))
            return -float(v[0]), -g[0]
        ref = scipy.optimize.minimize(neg_f, starts[j], jac=True, bounds=bounds, method="L-BFGS-B", options=dict(maxiter=15000))
        ev_ref += count[0]
        assert (st[j] == 1) == bool(ref.success), (j, st[j], ref.message)
        scale = max(1.0, abs(ref.fun))
        assert abs(-vs[j] - ref.fun) <= 1e-7 * scale, (j, vs[j], ref.fun)
        if np.max(np.abs(xs[j] - ref.x)) <= 1e-6:
            n_same += 1
    assert n_same >= R - 2, n_same                  # (flat directions of PI / EI far from the data: equal value, another point)
    assert 0.7 * ev_ref - 10 <= evals <= 1.4 * ev_ref + 10, (evals, ev_ref)
    b = ta.Bounds([("x%d" % d, 0.0, 1.0) for d in range(D)])
    got = {}
    for mode in (True, "scipy"):
        np.random.seed(5)
        x, info = ta.CandidateSweep(num_random=2000, grad_restarts=6, start_from_best=2, lockstep=mode)(b, f)
        got[mode] = (x, info["max_acq"])
    assert abs(got[True][1] - got["scipy"][1]) <= 1e-7 * max(1.0, abs(got["scipy"][1]))
    sur.close()


def test_acq_lbfgsb_argument_checks():
    import turbo_amd as ta
    X, y, _ = _synth(1, 40, 2, 1)
    gp = ta.NativeGP(0, "f64")
    gp.fit(X, y, "rbf", 1.0, 0.5, 1e-3, 1e-10, True)
    with pytest.raises(ValueError):
        gp.acq_refine(np.zeros((2, 2)), [1, 1], [0, 0], ta._lib.ACQ_EI, lbfgsb=True)      # lo > hi
    x, v, st, ev = gp.acq_refine(np.array([[0.2, 0.7], [5.0, -3.0]]), [0, 0], [1, 1], ta._lib.ACQ_UCB, -1.0, 0.0, 2.0, lbfgsb=True)
    assert np.all(x >= 0) and np.all(x <= 1) and ev >= 2 and set(st.tolist()) <= {0, 1, 2}
    vv, _ = gp.acq_grad(x, ta._lib.ACQ_UCB, -1.0, 0.0, 2.0)
    np.testing.assert_allclose(vv, v, rtol=1e-12, atol=1e-12)
    # one iteration only: stopped by max_iter (status 0) unless already stationary
    x1, v1, st1, ev1 = gp.acq_refine(np.array([[0.2, 0.7]]), [0, 0], [1, 1], ta._lib.ACQ_UCB, -1.0, 0.0, 2.0, max_iter=1, lbfgsb=True)
    assert st1[0] in (0, 1) and ev1 <= 22


def test_the_default_hyper_parameter_fit_never_calls_scipy(monkeypatch):
    """every kernel with bounded theta -- all hyper-parameters free, one of them fixed, no noise term at all, isotropic
    or ARD, the small path or the blocked one -- is optimised inside the library: scipy.optimize.minimize is not called"""
    import warnings
    import scipy.optimize
    import turbo_amd as ta

    def boom(*a, **k):
        raise AssertionError("scipy.optimize.minimize was called")
    monkeypatch.setattr(scipy.optimize, "minimize", boom)
    for N in (40, 200):
        X, y, _ = _synth(N, N, 3, 1)
        for k in (ta.GPKernel("matern52", 1.0, 0.8, 1e-2), ta.GPKernel("rbf", 1.0, np.ones(3), 1e-2, bounds={"noise": "fixed"}),
                  ta.GPKernel("matern32", 1.0, 0.8, 1e-2, bounds={"length_scale": "fixed"}), ta.GPKernel("matern52", 1.0, 0.8, None),
                  ta.GPKernel("rbf", 1.0, np.ones(3), None, bounds={"constant": "fixed"})):
            sur = ta.HipGPSurrogate(model_params=dict(kernel=k, normalize_y=True, random_state=0, alpha=1e-6), training_iterations=2)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                model, info = sur.construct_model(0, X, y)
            assert info["lml_evaluations"] >= 2 and np.isfinite(model.get_log_likelihood())
            assert len(model.get_hyper_params()) == len(k.theta)
            sur.close()


def test_two_factories_fitting_hyper_parameters_at_the_same_time():
    """two host threads, a factory each, both inside an optimised construct_model at once: the device's worker pool
    serves one fit after the other (tgp_workers_acquire holds the pool for the duration of a fit) -- no deadlock, and
    each factory gets what it gets alone"""
    import threading
    import warnings
    import turbo_amd as ta
    data = [_synth(40 + i, 300 + 40 * i, 3, 1)[:2] for i in range(2)]

    def make():
        return ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 0.8, 1e-2), normalize_y=True, random_state=0),
                                 training_iterations=3, param_continuity=False)

    def fit(sur, X, y):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            m, info = sur.construct_model(0, X, y)
        return m.get_log_likelihood(), m.get_hyper_params().tobytes(), info["lml_evaluations"]
    alone = []
    for X, y in data:
        s = make()
        alone.append(fit(s, X, y))
        s.close()
    surs = [make(), make()]
    out = [[], []]
    errs = []

    def work(i):
        try:
            for _ in range(3):
                out[i].append(fit(surs[i], *data[i]))
        except BaseException as e:      # noqa: B902 -- reported below
            errs.append(e)
    ths = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ths), "a hyper-parameter fit is stuck"
    assert not errs, errs
    for i in range(2):
        assert out[i] == [alone[i]] * 3
        surs[i].close()
