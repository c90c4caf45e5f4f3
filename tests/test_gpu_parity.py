"""GPU parity: the HIP path (through the C-ABI / plugin classes) against the golden vectors
generated from the reference and against the CPU oracle.  Run with ``-m gpu`` on an MI355X."""
import os
import pickle
import warnings

import numpy as np
import pytest

from conftest import golden_path
from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu

RTOL = 1e-5          # BASELINE.json north_star: 1e-5 rtol (fp64) on mean / variance / acquisition
VAR_ATOL = 1e-9      # x (c + noise) * y_std^2 : variance cancels near observed points
# fp32 sweep (the 1e-5 target is stated for fp64 only): about 7x the deviation measured at the
# BASELINE sizes (7e-5 / 4e-6 at C3), so a 10x regression fails
F32_MU_TOL = 5e-4    # x y_std
F32_VAR_TOL = 5e-5   # x (c + noise) * y_std^2

ACQS = {"ei": ("EI", 0.01), "pi": ("PI", 0.01), "ucb2": ("UCB", 2.0), "ucbinf": ("UCB", float("inf"))}


@pytest.fixture(scope="module")
def ta():
    import turbo_amd
    return turbo_amd


def _kernel(ta, c):
    ls = c["length_scale"]
    ls = float(ls[0]) if len(ls) == 1 else ls
    noise = float(c["noise"])
    return ta.GPKernel(str(c["kind"]), float(c["constant"]), ls, noise if noise > 0 else None)


def _surrogate(ta, c, dtype="f64"):
    return ta.HipGPSurrogate(model_params=dict(kernel=_kernel(ta, c), optimizer=None,
                                               normalize_y=bool(c["normalize_y"]),
                                               alpha=float(c["jitter"])),
                             training_iterations=1, dtype=dtype)


def _var_atol(c):
    return VAR_ATOL * (float(c["constant"]) + float(c["noise"])) * float(c["y_std"]) ** 2


def test_fit_state(ta, golden_case):
    c = golden_case
    sur = _surrogate(ta, c)
    model, info = sur.construct_model(0, c["X"], c["y"])
    assert info["iterations"] == 1
    ctx = sur._context()
    L = ctx.debug_read(ta._lib.BUF_L)
    alpha = ctx.debug_read(ta._lib.BUF_ALPHA)
    Linv = ctx.debug_read(ta._lib.BUF_LINV)
    np.testing.assert_allclose(L, c["L"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(L @ Linv, np.eye(L.shape[0]), rtol=0, atol=1e-8)
    np.testing.assert_allclose(alpha, c["alpha"], rtol=1e-6, atol=1e-8 * np.abs(c["alpha"]).max())
    assert model.get_log_likelihood() == pytest.approx(float(c["lml"]), rel=1e-9, abs=1e-8)
    assert model.y_mean == pytest.approx(float(c["y_mean"]), rel=1e-13, abs=1e-14)
    assert model.y_std == pytest.approx(float(c["y_std"]), rel=1e-13)
    np.testing.assert_allclose(model.get_hyper_params(), c["hyper_params"], rtol=1e-14, atol=0)  # sklearn round-trips theta through log/exp
    # (with and without a WhiteKernel term: turbo/modules/surrogates.py:350-362)
    assert model.get_hyper_param_names() == [str(s) for s in c["hyper_param_names"]]


def test_predict(ta, golden_case):
    c = golden_case
    model, _ = _surrogate(ta, c).construct_model(0, c["X"], c["y"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        mus, sig = model.predict(c["Xc"], return_std_dev=True)
        only = model.predict(c["Xc"])
    assert mus.shape == sig.shape == (c["Xc"].shape[0],)
    np.testing.assert_allclose(mus, c["mus"], rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(sig ** 2, c["sigmas"] ** 2, rtol=RTOL, atol=_var_atol(c))
    np.testing.assert_array_equal(only, mus)
    # a single point and a non-contiguous view (plots / 1-point calls: SURVEY 3.2)
    m1, s1 = model.predict(c["Xc"][3:4], return_std_dev=True)
    np.testing.assert_allclose(m1, mus[3:4], rtol=1e-12, atol=1e-12)
    mv = model.predict(c["Xc"][::2])
    np.testing.assert_allclose(mv, mus[::2], rtol=1e-12, atol=1e-12)


@pytest.mark.parametrize("ext", ["min", "max"])
@pytest.mark.parametrize("acq", list(ACQS))
def test_acquisition(ta, golden_case, acq, ext):
    c = golden_case
    model, _ = _surrogate(ta, c).construct_model(0, c["X"], c["y"])
    cls, param = ACQS[acq]
    fac = getattr(ta, cls)(param)
    args = [0, model, ext]
    if fac.get_type() == "improvement":
        args.append(float(c["incumbent_" + ext]))
    f, info = fac.construct_function(*args)
    assert f.get_name() == str(c["name_%s_%s" % (acq, ext)])
    assert info == ({"beta": param} if cls == "UCB" else {"xi": param})
    got = f(c["Xc"])
    want = c["acq_%s_%s" % (acq, ext)]
    assert got.shape == want.shape
    # EI/PI/UCB are smooth in (mu, sigma); sigma itself carries the variance cancellation floor,
    # so compare away from clamped points with rtol and everywhere with the propagated floor
    s_floor = np.sqrt(_var_atol(c))
    scale = max(1.0, float(np.abs(want).max()))
    well = c["sigmas"] > 100 * s_floor
    np.testing.assert_allclose(got[well], want[well], rtol=RTOL, atol=1e-9 * scale)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=(abs(param) if np.isfinite(param) else 1.0) * 2 * s_floor + 2 * s_floor + 1e-9 * scale)
    # arg-max agrees with the returned vector (lowest index on ties)
    bi, bv = f.maximise(c["Xc"])
    assert bi == int(np.argmax(got)) and bv == got[bi]


def test_branin_trace_config0(ta):
    """Config 0 through the plugin classes, replaying the candidate batches the reference drew."""
    with np.load(golden_path("branin_trace"), allow_pickle=False) as z:
        t = {k: z[k] for k in z.files}
    bounds = ta.Bounds([("x", -5.0, 10.0), ("y", 0.0, 15.0)])
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 1.0, 1.0),
                                              optimizer=None, normalize_y=True), training_iterations=1)
    acq_fac = ta.EI(xi=float(t["xi"]))
    xs, ys = t["trial_xs"], t["trial_ys"]
    for i, trial in enumerate(t["trials"]):
        X, y = xs[:trial], ys[:trial]
        model, info = sur.construct_model(int(trial), X, y)
        assert model.get_log_likelihood() == pytest.approx(float(t["lml"][i]), rel=1e-9)
        f, _ = acq_fac.construct_function(int(trial), model, "min", float(y.min()))
        cand = t["cand_%d" % trial]
        aux = ta.CandidateSweep(num_random=cand.shape[0], gen_random=lambda n, lb, c=cand: c)
        x, minfo = aux(bounds, f)
        assert x.shape == (1, 2)
        assert minfo["max_acq"] == pytest.approx(float(t["max_acq"][i]), rel=RTOL)
        np.testing.assert_array_equal(x.reshape(-1), t["sel_x"][i])


def test_not_pd_raises_linalgerror(ta):
    with np.load(golden_path("not_pd"), allow_pickle=False) as z:
        sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("rbf", 1.0, 0.5, None),
                                                  optimizer=None, normalize_y=True, alpha=0.0),
                                training_iterations=1)
        with pytest.raises(np.linalg.LinAlgError):
            sur.construct_model(0, z["X"], z["y"])
        # the context stays usable afterwards
        sur.model_params["alpha"] = 1e-6
        model, _ = sur.construct_model(1, z["X"], z["y"])
        assert np.isfinite(model.get_log_likelihood())


def test_bad_arguments(ta):
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("rbf", 1.0, 0.5, 1e-3), optimizer=None),
                            training_iterations=1)
    X = np.random.RandomState(0).rand(5, 3)
    model, _ = sur.construct_model(0, X, np.arange(5.0))
    with pytest.raises(AssertionError):
        model.predict(np.zeros((4, 2)))
    with pytest.raises(AssertionError):
        sur.construct_model(0, X, np.arange(4.0))
    with pytest.raises(ValueError):
        ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel(), optimizer="simplex"),
                          training_iterations=2).construct_model(0, X, np.arange(5.0))
    with pytest.raises(AssertionError):
        ta.EI(0.01).construct_function(0, object(), "min", 0.0)      # no predict(): not a model at all


def test_empty_and_one_dimensional_inputs(ta):
    X = np.random.RandomState(0).rand(9, 3)
    model, _ = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("rbf", 1.0, 0.5, 1e-3), optimizer=None),
                                 training_iterations=1).construct_model(0, X, np.arange(9.0))
    mu, sg = model.predict(np.empty((0, 3)), return_std_dev=True)
    assert mu.shape == sg.shape == (0,)
    f, _ = ta.UCB(1.0).construct_function(0, model, "max")
    assert f(np.empty((0, 3))).shape == (0,)
    np.testing.assert_array_equal(model.predict(X[4]), model.predict(X[4:5]))     # a bare (D,) point


def test_config4_shapes(ta):
    """64D Matern-3/2, N=8192, PI, fp32 sweep (BASELINE config 4) on a 65 536-candidate shard:
    properties plus the oracle on a bounded sample"""
    N, D, M = 8192, 64, 65536
    X, y, _ = _synth(1004, N, D, 1)
    ls, noise = float(np.sqrt(D / 6.0)), 1e-2
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern32", 1.0, ls, noise), optimizer=None,
                                              normalize_y=True), training_iterations=1, dtype="f32")
    model, _ = sur.construct_model(0, X, y)
    Xc = np.random.RandomState(3004).uniform(0, 1, size=(M, D))
    f, _ = ta.PI(0.01).construct_function(0, model, "min", float(y.min()))
    full = f(Xc)
    assert full.shape == (M,) and np.all((full >= 0) & (full <= 1))
    h = M // 2
    np.testing.assert_array_equal(np.concatenate([f(Xc[:h]), f(Xc[h:])]), full)
    bi, bv = f.maximise(Xc)
    assert bi == int(np.argmax(full)) and bv == full[bi]
    om = o.fit(X, y, "matern32", 1.0, ls, noise, 1e-10, True)
    assert model.get_log_likelihood() == pytest.approx(om.lml, rel=1e-8)
    mu, sg = model.predict(Xc[:1024], return_std_dev=True)
    omu, osig = o.predict(om, Xc[:1024])
    assert np.max(np.abs(mu - omu)) < F32_MU_TOL * om.y_std
    assert np.max(np.abs(sg ** 2 - osig ** 2)) < F32_VAR_TOL * (1 + noise) * om.y_std ** 2


def test_model_survives_pickle_and_eviction(ta):
    """Recorder keeps one model per trial and pickles them (turbo/recorder.py:117-155)."""
    with np.load(golden_path("rbf_iso_8d"), allow_pickle=False) as z:
        c = {k: z[k] for k in z.files}
    sur = _surrogate(ta, c)
    m1, _ = sur.construct_model(0, c["X"], c["y"])
    ref = m1.predict(c["Xc"])
    m2, _ = sur.construct_model(1, c["X"][:20], c["y"][:20])   # displaces m1 in the context
    assert m2.predict(c["Xc"]).shape == ref.shape
    np.testing.assert_array_equal(m1.predict(c["Xc"]), ref)     # refit on demand, bit-identical
    m3 = pickle.loads(pickle.dumps(m1))
    np.testing.assert_array_equal(m3.predict(c["Xc"]), ref)
    assert not m3._factory._context().host          # with a GPU in sight a reloaded model runs on it
    try:
        import dill
        m4 = dill.loads(dill.dumps(m1))
        np.testing.assert_array_equal(m4.predict(c["Xc"]), ref)
    except ImportError:
        pass


@pytest.mark.parametrize("N,kind", [(40, "matern52"), (600, "rbf")])
def test_model_pickled_here_reloads_in_a_process_without_gpu(ta, tmp_path, N, kind):
    """SURVEY 8b "Pickling" / 8f-4: a model fitted on the GPU, pickled (Recorder.save_compressed,
    turbo/recorder.py:117-155), unpickled by a process that sees NO HIP device (the plot path,
    turbo/recorder.py:157-163, turbo/plotting/trials.py:574-577) answers from libturbogp.so's host
    backend with the GPU's values; and a state blob exported on the GPU imports on a host handle."""
    import subprocess
    import sys
    X, y, Xc = _synth(N, N, 5, 700)
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel(kind, 1.4, 0.8, 1e-3), optimizer=None, normalize_y=True),
                            training_iterations=1)
    model, _ = sur.construct_model(0, X, y)
    assert not sur._context().host
    mu, sg = model.predict(Xc, return_std_dev=True)
    ei, _ = ta.EI(0.01).construct_function(0, model, "min", float(y.min()))
    acq = ei(Xc)
    blob = sur._context().export_state()
    pk, xc, out = tmp_path / "m.pkl", tmp_path / "xc.npy", tmp_path / "out.npz"
    pk.write_bytes(pickle.dumps(model))
    np.save(xc, Xc)
    child = (
        "import sys, pickle, warnings, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import turbo_amd as ta\n"
        "warnings.simplefilter('ignore')\n"
        "m = pickle.loads(open(sys.argv[1], 'rb').read())\n"
        "Xc = np.load(sys.argv[2])\n"
        "mu, sg = m.predict(Xc, return_std_dev=True)\n"
        "assert m._factory._context().host\n"
        "f, _ = ta.EI(0.01).construct_function(0, m, 'min', %r)\n"
        "np.savez(sys.argv[3], mu=mu, sg=sg, acq=f(Xc), lml=np.array(m.get_log_likelihood()))\n"
        "print('host-reload ok')\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), float(y.min())))
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    res = subprocess.run([sys.executable, "-c", child, str(pk), str(xc), str(out)], env=env, capture_output=True,
                         text=True, timeout=300)
    assert res.returncode == 0 and "host-reload ok" in res.stdout, res.stdout[-2000:] + res.stderr[-4000:]
    with np.load(out, allow_pickle=False) as z:
        np.testing.assert_allclose(z["mu"], mu, rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(z["sg"] ** 2, sg ** 2, rtol=1e-7, atol=1e-9 * 1.401 * model.y_std ** 2)
        np.testing.assert_allclose(z["acq"], acq, rtol=1e-5, atol=1e-7)
        assert float(z["lml"]) == pytest.approx(model.get_log_likelihood(), rel=1e-10)
    host = ta._lib.NativeGP(ta._lib.DEVICE_HOST, "f64")
    assert host.import_state(blob) == pytest.approx(model.get_log_likelihood(), rel=1e-10)
    r = host.evaluate(Xc, ta._lib.ACQ_NONE, want_mu=True, want_sigma=True)
    np.testing.assert_allclose(r["mu"], mu, rtol=1e-9, atol=1e-10)


def _synth(seed, N, D, M):
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 1, size=(N, D))
    w = rng.normal(size=D) / np.sqrt(D)
    y = np.sin(3 * X @ w) + 0.5 * ((X - 0.5) ** 2).sum(1) + 0.01 * rng.normal(size=N)
    Xc = rng.uniform(0, 1, size=(M, D))
    return X, y, Xc


@pytest.mark.parametrize("kind,N,D,M,ard", [("rbf", 700, 8, 5000, False),
                                            ("matern52", 1024, 16, 40000, True),
                                            ("matern32", 300, 64, 1000, False)])
def test_midsize_vs_oracle(ta, kind, N, D, M, ard):
    """ragged N and M, several tiles, several launches (M > chunk) against the CPU oracle"""
    X, y, Xc = _synth(11 + N, N, D, M)
    ls = np.sqrt(D / 6.0) * ((0.5 + np.arange(D) / (D - 1.0)) if ard else 1.0)
    noise = 1e-4
    model, _ = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel(kind, 1.0, ls, noise),
                                                   optimizer=None, normalize_y=True),
                                 training_iterations=1).construct_model(0, X, y)
    om = o.fit(X, y, kind, 1.0, ls, noise, 1e-10, True)
    assert model.get_log_likelihood() == pytest.approx(om.lml, rel=1e-9)
    mus, sig = model.predict(Xc, return_std_dev=True)
    omu, osig = o.predict(om, Xc, chunk=8192)
    np.testing.assert_allclose(mus, omu, rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(sig ** 2, osig ** 2, rtol=RTOL, atol=VAR_ATOL * (1 + noise) * om.y_std ** 2)
    f, _ = ta.EI(0.01).construct_function(0, model, "min", float(y.min()))
    got = f(Xc)
    want = o.acquisition("ei", omu, osig, "min", 0.01, float(y.min()))
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=1e-12)
    bi, bv = f.maximise(Xc)
    assert bi == int(np.argmax(want))


@pytest.mark.parametrize("N", [513, 520, 767, 769, 1025, 1279, 1281, 1537, 2305, 3073, 3329])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_outer_block_boundaries_vs_oracle(ta, N, dtype):
    """sizes on either side of the fit's outer-block boundaries (256 / 512 columns, a last block of
    256, padding rows inside the last panel): the inverse factor is assembled block row by block
    row on the background stream there, with its products, z and alpha shares per block -- L,
    alpha, the LML, the mean and the variance against the oracle, for an f64 and an f32 sweep"""
    D, M = 5, 3000
    X, y, Xc = _synth(900 + N, N, D, M)
    ls, noise = np.linspace(0.6, 1.2, D), 1e-3
    gp = ta.NativeGP(0, dtype)
    lml, ym, ys = gp.fit(X, y, "matern52", 1.2, ls, noise, 1e-10, True)
    om = o.fit(X, y, "matern52", 1.2, ls, noise, 1e-10, True)
    assert lml == pytest.approx(om.lml, rel=1e-9)
    np.testing.assert_allclose(np.tril(gp.debug_read(ta._lib.BUF_L)), om.L, rtol=1e-8, atol=1e-11)
    np.testing.assert_allclose(gp.debug_read(ta._lib.BUF_ALPHA), om.alpha, rtol=1e-6, atol=1e-8 * np.abs(om.alpha).max())
    gp.set_candidates(Xc)
    r = gp.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)
    mu, sg = o.predict(om, Xc)
    if dtype == "f64":
        np.testing.assert_allclose(r["mu"], mu, rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(r["sigma"] ** 2, sg ** 2, rtol=RTOL, atol=VAR_ATOL * (1.2 + noise) * om.y_std ** 2)
    else:
        assert np.max(np.abs(r["mu"] - mu)) <= F32_MU_TOL * om.y_std
        assert np.max(np.abs(r["sigma"] ** 2 - sg ** 2)) <= F32_VAR_TOL * (1.2 + noise) * om.y_std ** 2
    assert r["best_idx"] == int(np.argmax(r["acq"]))


def test_f32_sweep_accuracy(ta):
    """fp32 sweep (configs 3/4): f64 fit, f32 cross-kernel + contraction.  1e-5 is an fp64 target;
    here the deviation from the f64 oracle is bounded and the arg-max regret is checked."""
    N, D, M = 1024, 32, 8192
    X, y, Xc = _synth(5, N, D, M)
    ls, noise = float(np.sqrt(D / 6.0)), 1e-2
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("rbf", 1.0, ls, noise), optimizer=None,
                                              normalize_y=True), training_iterations=1, dtype="f32")
    model, _ = sur.construct_model(0, X, y)
    om = o.fit(X, y, "rbf", 1.0, ls, noise, 1e-10, True)
    assert model.get_log_likelihood() == pytest.approx(om.lml, rel=1e-9)   # fit is f64
    mus, sig = model.predict(Xc, return_std_dev=True)
    omu, osig = o.predict(om, Xc)
    assert np.max(np.abs(mus - omu)) < F32_MU_TOL * om.y_std
    assert np.max(np.abs(sig ** 2 - osig ** 2)) < F32_VAR_TOL * (1 + noise) * om.y_std ** 2
    f, _ = ta.EI(0.01).construct_function(0, model, "min", float(y.min()))
    want = o.acquisition("ei", omu, osig, "min", 0.01, float(y.min()))
    bi, _ = f.maximise(Xc)
    assert want[bi] >= want.max() - 1e-3 * max(want.max(), 1e-12) - 1e-9   # regret of the f32 choice


@pytest.mark.parametrize("kind,N,D,M,acq", [("rbf", 700, 8, 5000, "ei"), ("matern52", 2304, 9, 6007, "ucb"),
                                            ("matern32", 1300, 64, 3001, "pi"), ("rbf", 4096, 32, 40000, "ei"),
                                            ("matern12", 260, 3, 777, "ei")])
@pytest.mark.parametrize("split", ["f32x3", "f32h2"])
def test_f32x3_sweep_is_f32_accurate(ta, kind, N, D, M, acq, split):
    """dtypes 'f32x3' / 'f32h2' (opt-in): the contraction on the bf16 / fp16 matrix pipe from three
    bf16 planes (six products) or two scaled fp16 planes (three products) per f32 operand,
    accumulated in f32 (csrc/trmm_bf16x3.hpp, trmm_f16x2.hpp).  It has to meet the SAME bounds
    against the f64 oracle as the f32 sweep, pick the same candidate as the f32 sweep would, and be
    shard-invariant bit for bit; ragged N and M, every kernel family, every acquisition."""
    X, y, Xc = _synth(40 + N, N, D, M)
    ls, noise = float(np.sqrt(D / 6.0)), 1e-2
    om = o.fit(X, y, kind, 1.0, ls, noise, 1e-10, True)
    nchk = min(M, 6000)
    omu, osig = o.predict(om, Xc[:nchk])
    code = {"ei": ta._lib.ACQ_EI, "ucb": ta._lib.ACQ_UCB, "pi": ta._lib.ACQ_PI}[acq]
    param = 2.0 if acq == "ucb" else 0.01
    res = {}
    for dt in ("f32", split):
        gp = ta.NativeGP(0, dt)
        lml, _, _ = gp.fit(X, y, kind, 1.0, ls, noise, 1e-10, True)
        assert lml == pytest.approx(om.lml, rel=1e-9)
        gp.set_candidates(Xc)
        r = gp.sweep(code, -1.0, float(y.min()), param, want_mu=True, want_sigma=True, want_acq=True)
        assert np.max(np.abs(r["mu"][:nchk] - omu)) < F32_MU_TOL * om.y_std, dt
        assert np.max(np.abs(r["sigma"][:nchk] ** 2 - osig ** 2)) < F32_VAR_TOL * (1 + noise) * om.y_std ** 2, dt
        assert r["best_idx"] == int(np.argmax(r["acq"]))
        res[dt] = r
        if dt == split:
            # shards of the batch give the same rows, bit for bit
            cut = (M // 3 // 7) * 7 + 5
            gp.set_candidates(Xc[:cut])
            r1 = gp.sweep(code, -1.0, float(y.min()), param, want_mu=True, want_sigma=True)
            gp.set_candidates(Xc[cut:])
            r2 = gp.sweep(code, -1.0, float(y.min()), param, want_mu=True, want_sigma=True)
            np.testing.assert_array_equal(np.concatenate([r1["mu"], r2["mu"]]), r["mu"])
            np.testing.assert_array_equal(np.concatenate([r1["sigma"], r2["sigma"]]), r["sigma"])
            # a second fit on the same handle cuts new planes
            gp.fit(X[: N - 3], y[: N - 3], kind, 1.0, ls, noise, 1e-10, True)
            gp.set_candidates(Xc[:500])
            r3 = gp.sweep(code, -1.0, float(y.min()), param, want_mu=True, want_sigma=True)
            om3 = o.fit(X[: N - 3], y[: N - 3], kind, 1.0, ls, noise, 1e-10, True)
            mu3, sg3 = o.predict(om3, Xc[:500])
            assert np.max(np.abs(r3["sigma"] ** 2 - sg3 ** 2)) < F32_VAR_TOL * (1 + noise) * om3.y_std ** 2
            # ... and so does a one-row append (tgp_fit_append), and the one-call / top-k entries use the planes too
            gp.fit(X[: N - 2], y[: N - 2], kind, 1.0, ls, noise, 1e-10, True, append=True)
            assert gp.appended
            om4 = o.fit(X[: N - 2], y[: N - 2], kind, 1.0, ls, noise, 1e-10, True)
            mu4, sg4 = o.predict(om4, Xc[:500])
            r4 = gp.evaluate(Xc[:500], code, -1.0, float(y.min()), param, want_mu=True, want_sigma=True, want_acq=True)
            assert np.max(np.abs(r4["mu"] - mu4)) < F32_MU_TOL * om4.y_std
            assert np.max(np.abs(r4["sigma"] ** 2 - sg4 ** 2)) < F32_VAR_TOL * (1 + noise) * om4.y_std ** 2
            idx, vals = gp.sweep_topk(5, code, -1.0, float(y.min()), param)
            np.testing.assert_array_equal(idx, np.argsort(-r4["acq"], kind="stable")[:5])
    # the variance of the split path is as close to the oracle as the f32 path's (2x slack)
    e32 = np.max(np.abs(res["f32"]["sigma"][:nchk] ** 2 - osig ** 2))
    e3 = np.max(np.abs(res[split]["sigma"][:nchk] ** 2 - osig ** 2))
    assert e3 <= 2.0 * e32 + 1e-7 * om.y_std ** 2, (e3, e32)
    want = o.acquisition(acq, omu, osig, "min", param, float(y.min()))
    if res[split]["best_idx"] < nchk:
        bi = res[split]["best_idx"]
        assert want[bi] >= want.max() - 1e-3 * max(abs(want.max()), 1e-12) - 1e-9


# ---- BASELINE.json full sizes: size-independent properties ---------------------------------

def _full_model(ta, N, D, kind, dtype, noise, seed):
    X, y, _ = _synth(seed, N, D, 1)
    ls = float(np.sqrt(D / 6.0))
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel(kind, 1.0, ls, noise), optimizer=None,
                                              normalize_y=True), training_iterations=1, dtype=dtype)
    model, _ = sur.construct_model(0, X, y)
    return sur, model, X, y


@pytest.mark.parametrize("N,D,M,kind,dtype,noise", [(512, 8, 65536, "rbf", "f64", 1e-4),
                                                    (4096, 32, 262144, "rbf", "f32", 1e-2)])
def test_full_size_properties(ta, N, D, M, kind, dtype, noise):
    sur, model, X, y = _full_model(ta, N, D, kind, dtype, noise, 1000 + N)
    rng = np.random.RandomState(3000 + N)
    Xc = rng.uniform(0, 1, size=(M, D))
    f, _ = ta.EI(0.01).construct_function(0, model, "min", float(y.min()))
    full = f(Xc)
    assert full.shape == (M,) and np.all(np.isfinite(full)) and np.all(full >= -1e-12)
    # (1) sharding invariance: two halves == the whole (rows are independent)
    h = M // 2
    np.testing.assert_array_equal(np.concatenate([f(Xc[:h]), f(Xc[h:])]), full)
    # (2) arg-max == arg-max of the vector, and idempotence
    bi, bv = f.maximise(Xc)
    assert bi == int(np.argmax(full)) and bv == full[bi]
    assert f.maximise(Xc) == (bi, bv)
    # (3) permutation equivariance on a slice
    perm = rng.permutation(4096)
    np.testing.assert_array_equal(f(Xc[:4096][perm]), full[:4096][perm])
    # (4) the mean interpolates: mu(x_i) = y_std*(yn_i - (noise+jitter)*alpha_i) + y_mean
    alpha = sur._context().debug_read(ta._lib.BUF_ALPHA)
    mu_train = model.predict(X)
    yn = (y - model.y_mean) / model.y_std
    expect = model.y_std * (yn - (noise + 1e-10) * alpha) + model.y_mean
    tol = 1e-7 if dtype == "f64" else F32_MU_TOL
    np.testing.assert_allclose(mu_train, expect, rtol=0, atol=tol * model.y_std)
    # (5) UCB is linear in beta: ucb(b) = -mu + b*sigma  (minimising)
    mu, sg = model.predict(Xc[:8192], return_std_dev=True)
    u, _ = ta.UCB(2.0).construct_function(0, model, "min")
    np.testing.assert_allclose(u(Xc[:8192]), -mu + 2.0 * sg, rtol=1e-14, atol=1e-14)
    # (6) against the oracle on a bounded sample
    om = o.fit(X, y, kind, 1.0, float(np.sqrt(D / 6.0)), noise, 1e-10, True)
    omu, osig = o.predict(om, Xc[:2048])
    if dtype == "f64":
        np.testing.assert_allclose(mu[:2048], omu, rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(sg[:2048] ** 2, osig ** 2, rtol=RTOL, atol=VAR_ATOL * (1 + noise) * om.y_std ** 2)
    else:
        assert np.max(np.abs(mu[:2048] - omu)) < F32_MU_TOL * om.y_std
        assert np.max(np.abs(sg[:2048] ** 2 - osig ** 2)) < F32_VAR_TOL * (1 + noise) * om.y_std ** 2


# ---- "next" row SURVEY 8(f)1: LML gradient and hyper-parameter optimisation on the GPU -----------

GRAD_CASES = ["grad_rbf_iso_3d", "grad_rbf_ard_3d", "grad_matern52_ard_5d", "grad_matern32_iso_5d",
              "grad_matern12_iso_5d_nowhite", "grad_default_matern52_white"]


@pytest.mark.parametrize("name", GRAD_CASES)
def test_lml_gradient(ta, name):
    with np.load(golden_path(name), allow_pickle=False) as z:
        c = {k: z[k] for k in z.files}
    noise = float(c["noise"])
    ls = c["length_scale"]
    gp = ta.NativeGP(0, "f64")
    lml, grad = gp.fit_grad(c["X"], c["y"], str(c["kind"]), float(c["constant"]), ls,
                            max(noise, 0.0), float(c["jitter"]), True)
    assert lml == pytest.approx(float(c["lml"]), rel=1e-9)
    want = c["grad"]
    got = grad if noise >= 0 else grad[:-1]
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-8)
    # the model stays usable for sweeps at these hyper-parameters
    gp.set_candidates(c["X"][:5])
    r = gp.sweep(want_mu=True)
    assert np.all(np.isfinite(r["mu"]))


def test_lml_gradient_larger_vs_oracle(ta):
    X, y, _ = _synth(77, 700, 6, 1)
    ls = np.linspace(0.6, 1.4, 6)
    gp = ta.NativeGP(0, "f64")
    lml, grad = gp.fit_grad(X, y, "matern52", 1.7, ls, 3e-3, 1e-10, True)
    olml, ograd = o.lml_and_grad(X, y, "matern52", 1.7, ls, 3e-3, 1e-10, True)
    assert lml == pytest.approx(olml, rel=1e-9)
    np.testing.assert_allclose(grad, ograd, rtol=1e-6, atol=1e-6)
    lml, grad = gp.fit_grad(X, y, "rbf", 0.9, 0.8, 1e-2, 1e-10, True)
    olml, ograd = o.lml_and_grad(X, y, "rbf", 0.9, 0.8, 1e-2, 1e-10, True)
    np.testing.assert_allclose(grad, ograd, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("N,D", [(5, 1), (20, 3), (64, 16), (65, 2), (100, 17), (128, 64), (128, 5), (97, 65)])
@pytest.mark.parametrize("kind", ["rbf", "matern52", "matern32", "matern12"])
def test_lml_gradient_small_problems_vs_oracle(ta, N, D, kind):
    """N <= 128 (and D <= 64): fit and gradient are two one-workgroup launches (small_kernels.hip,
    grad_kernels.hip small_grad_kernel), ARD and isotropic; (97, 65) is one dimension too wide and
    takes the blocked path -- same answers"""
    X, y, _ = _synth(500 + N + D, N, D, 1)
    gp = ta.NativeGP(0, "f64")
    ls_ard = np.linspace(0.7, 1.6, D) * np.sqrt(D / 6.0)
    for ls in (ls_ard, float(np.sqrt(D / 6.0))):
        lml, grad = gp.fit_grad(X, y, kind, 1.3, ls, 2e-3, 1e-10, True)
        olml, ograd = o.lml_and_grad(X, y, kind, 1.3, ls, 2e-3, 1e-10, True)
        assert lml == pytest.approx(olml, rel=1e-9, abs=1e-9)
        np.testing.assert_allclose(grad, ograd, rtol=1e-7, atol=1e-8 * max(1.0, float(np.abs(ograd).max())))
    # the fitted state that a gradient evaluation leaves behind predicts like a plain fit
    Xc = np.random.RandomState(3).uniform(0, 1, (50, D))
    gp.set_candidates(Xc)
    r = gp.sweep(want_mu=True, want_sigma=True)
    om = o.fit(X, y, kind, 1.3, float(np.sqrt(D / 6.0)), 2e-3, 1e-10, True)
    mu, sg = o.predict(om, Xc)
    np.testing.assert_allclose(r["mu"], mu, rtol=1e-8, atol=1e-9 * om.y_std)
    np.testing.assert_allclose(r["sigma"], sg, rtol=1e-6, atol=1e-9 * om.y_std)


@pytest.mark.parametrize("name,kernel_of", [
    ("opt_default_2d", lambda ta: ta.GPKernel("matern52", 1.0, 1.0, 1.0)),
    ("opt_rbf_ard_4d", lambda ta: ta.GPKernel("rbf", 1.0, np.ones(4), 1e-2)),
    # round 5 (tests/golden/make_golden_hyper2.py): sizes either side of the one-launch limit; fixed hyper-parameters
    ("opt_matern32_iso_5d_mid", lambda ta: ta.GPKernel("matern32", 1.0, 0.9, 1e-2)),
    ("opt_fixed_noise_3d", lambda ta: ta.GPKernel("matern52", 1.0, 0.8, 1e-2, bounds={"noise": "fixed"})),
    ("opt_fixed_constant_ard_3d", lambda ta: ta.GPKernel("rbf", 1.0, np.ones(3), 1e-2, bounds={"constant": "fixed"})),
    ("opt_nowhite_matern52_3d", lambda ta: ta.GPKernel("matern52", 1.0, 0.8, None)),        # no noise term, alpha = 1e-3
])
def test_hyper_parameter_optimisation_trace(ta, name, kernel_of):
    """the reference's default usage: training_iterations > 0, warm start across trials
    (param_continuity), iterations - 1 random restarts with random_state=0"""
    with np.load(golden_path(name), allow_pickle=False) as z:
        t = {k: z[k] for k in z.files}
    extra = dict(alpha=float(t["alpha"])) if "alpha" in t else {}
    sur = ta.HipGPSurrogate(model_params=dict(kernel=kernel_of(ta), normalize_y=True, random_state=0, **extra),
                            training_iterations=int(t["iters"]), param_continuity=True)
    np.testing.assert_allclose(kernel_of(ta).theta_bounds, t["bounds"], rtol=1e-12)
    for k, n in enumerate(t["sizes"]):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model, info = sur.construct_model(k, t["X"][:n], t["y"][:n])
        assert info["iterations"] == int(t["iters"]) and info["lml_evaluations"] > 0
        assert model.get_hyper_param_names() == [str(s) for s in t["names"]]
        # same optimum as the reference's sklearn path: likelihood to 1e-6, parameters to 1e-3
        assert model.get_log_likelihood() == pytest.approx(float(t["lml_%d" % k]), rel=1e-6, abs=1e-6)
        hp, want = model.get_hyper_params(), t["hp_%d" % k]
        free = (want > 1.1e-5) & (want < 0.9e5)       # not pinned to a bound
        np.testing.assert_allclose(np.log(hp[free]), np.log(want[free]), atol=2e-3)
        mu, sg = model.predict(t["X"][-16:], return_std_dev=True)
        np.testing.assert_allclose(mu, t["mu_%d" % k], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(sg, t["sigma_%d" % k], rtol=1e-3, atol=1e-5)


# optimizer='device' walks its own iterates (a projected L-BFGS in one launch), not SciPy's L-BFGS-B:
# on the reference's recorded traces it ends at scikit-learn's optimum in four of the five trials and
# at a BETTER local optimum in one.  Recorded here as an expectation, not as prose: the deviation is
# why the option stays opt-in.
DEVICE_OPT_KNOWN_BETTER = {("opt_rbf_ard_4d", 0): -17.77}     # (trace, trial) -> LML reached (scikit-learn: -36.18)


@pytest.mark.parametrize("name,kernel_of", [
    ("opt_default_2d", lambda ta: ta.GPKernel("matern52", 1.0, 1.0, 1.0)),
    ("opt_rbf_ard_4d", lambda ta: ta.GPKernel("rbf", 1.0, np.ones(4), 1e-2)),
])
def test_device_optimizer_on_the_reference_traces(ta, name, kernel_of):
    with np.load(golden_path(name), allow_pickle=False) as z:
        t = {k: z[k] for k in z.files}
    sur = ta.HipGPSurrogate(model_params=dict(kernel=kernel_of(ta), normalize_y=True, random_state=0, optimizer="device"),
                            training_iterations=int(t["iters"]), param_continuity=True)
    for k, n in enumerate(t["sizes"]):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model, info = sur.construct_model(k, t["X"][:n], t["y"][:n])
        ref = float(t["lml_%d" % k])
        got = model.get_log_likelihood()
        assert got >= ref - 1e-6 * abs(ref), (name, k, got, ref)            # never a worse optimum than the reference's
        if (name, k) in DEVICE_OPT_KNOWN_BETTER:
            assert got == pytest.approx(DEVICE_OPT_KNOWN_BETTER[(name, k)], abs=0.02), (name, k, got)
            assert got > ref + 1.0
        else:
            assert got == pytest.approx(ref, rel=1e-6, abs=1e-6), (name, k, got, ref)
            hp, want = model.get_hyper_params(), t["hp_%d" % k]
            free = (want > 1.1e-5) & (want < 0.9e5)
            np.testing.assert_allclose(np.log(hp[free]), np.log(want[free]), atol=2e-3)


@pytest.mark.parametrize("kind,N,D,ard", [("matern52", 10, 2, False), ("matern52", 32, 2, False), ("rbf", 64, 4, True),
                                          ("matern52", 65, 3, True), ("matern32", 128, 8, True),
                                          ("matern52", 128, 16, False)])
def test_one_launch_hyper_fit_vs_scipy_driven(ta, kind, N, D, ard):
    """optimizer='device' (tgp_fit_optimise: every start optimised in its own workgroup of ONE
    launch) against the default (SciPy's L-BFGS-B driving tgp_fit_grad, what scikit-learn does):
    from the same starts the likelihood reached is the same to 1e-6 or better, and where it is the
    same so are the hyper-parameters; the raw entry point's f is -LML at its theta, which is
    stationary inside the bounds"""
    rng = np.random.RandomState(N + D)
    X = rng.uniform(0, 1, (N, D))
    y = np.sin(3 * X @ rng.normal(size=D)) + 0.5 * ((X - 0.5) ** 2).sum(1) + 0.05 * rng.normal(size=N)
    got = {}
    for opt in ("fmin_l_bfgs_b", "device"):
        k = ta.GPKernel(kind, 1.0, np.ones(D) if ard else 1.0, 1e-2)
        sur = ta.HipGPSurrogate(model_params=dict(kernel=k, normalize_y=True, random_state=0, optimizer=opt),
                                training_iterations=3, param_continuity=False)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model, info = sur.construct_model(0, X, y)
        assert info["lml_evaluations"] > 0
        got[opt] = (model.get_log_likelihood(), np.log(model.get_hyper_params()))
    (l_ref, t_ref), (l_dev, t_dev) = got["fmin_l_bfgs_b"], got["device"]
    assert l_dev >= l_ref - 1e-6 * max(1.0, abs(l_ref)), (l_dev, l_ref)
    if abs(l_dev - l_ref) <= 1e-6 * max(1.0, abs(l_ref)):
        free = (t_ref > np.log(1.1e-5)) & (t_ref < np.log(0.9e5))
        np.testing.assert_allclose(t_dev[free], t_ref[free], atol=5e-3)
    # the raw entry point
    k = ta.GPKernel(kind, 1.0, np.ones(D) if ard else 1.0, 1e-2)
    b = k.theta_bounds
    n_ls = D if ard else 1
    starts = np.vstack([k.theta, np.random.RandomState(1).uniform(b[:, 0], b[:, 1], (3, len(b)))])
    gp = ta.NativeGP(0, "f64")
    theta, f, st, ev = gp.fit_optimise(X, y, kind, starts, n_ls, b, 1e-10, True)
    assert theta.shape == starts.shape and ev >= len(starts) and np.all((st == 1) | (st == 2))
    assert np.all(theta >= b[:, 0]) and np.all(theta <= b[:, 1])
    for s in range(len(starts)):
        if not np.isfinite(f[s]):
            continue
        p = np.exp(theta[s])
        lml, grad = gp.fit_grad(X, y, kind, p[0], p[1:1 + n_ls] if ard else p[1], p[-1], 1e-10, True)
        assert -f[s] == pytest.approx(lml, rel=1e-9, abs=1e-9)
        inner = (theta[s] > b[:, 0] + 1e-9) & (theta[s] < b[:, 1] - 1e-9)
        if st[s] == 1:
            assert np.max(np.abs(grad[inner]), initial=0.0) < 1e-2 * max(1.0, abs(lml))
    with pytest.raises(ValueError):
        gp.fit_optimise(X, y, kind, starts, n_ls, np.log(np.array([[1e5, 1e-5]] * len(b))), 1e-10, True)    # lo > hi


def test_one_launch_hyper_fit_degenerate_inputs(ta):
    """tgp_fit_optimise on inputs the optimiser can fail on: duplicated training points with the
    noise bounded to ~0 (kernel matrix not positive definite at every theta: each start ends
    with status 2 and f = inf, nothing hangs), constant targets without normalisation, and one
    start sitting on a corner of the bounds"""
    rng = np.random.RandomState(3)
    X = rng.uniform(0, 1, (40, 3))
    X[20:] = X[:20]                                            # every point twice
    y = np.sin(4 * X[:, 0]) + 0.1 * rng.normal(size=40)
    gp = ta.NativeGP(0, "f64")
    b = np.log(np.array([[1e-2, 1e2], [1e-2, 1e2], [1e-30, 1e-28]]))
    starts = np.vstack([b.mean(axis=1), b[:, 0], b[:, 1]])
    theta, f, st, ev = gp.fit_optimise(X, y, "rbf", starts, 1, b, 0.0, True, max_iter=50)
    assert np.all(st == 2) and np.all(np.isinf(f)) and ev == 3
    np.testing.assert_array_equal(theta, starts)               # a failed start stays where it began
    # the same data with a usable noise range: every start converges to a finite likelihood
    b[2] = np.log([1e-6, 1e1])
    starts = np.vstack([b.mean(axis=1), b[:, 0], b[:, 1]])
    theta, f, st, ev = gp.fit_optimise(X, y, "rbf", starts, 1, b, 1e-10, True)
    assert np.all(st == 1) and np.all(np.isfinite(f)) and np.all(theta >= b[:, 0]) and np.all(theta <= b[:, 1])
    # constant targets, normalize_y=False: alpha = K^-1 y is fine, the optimum pushes the constant up or down
    yc = np.full(40, 2.5)
    theta, f, st, ev = gp.fit_optimise(X[:20], yc[:20], "matern52", starts, 1, b, 1e-10, False)
    assert np.all((st == 1) | (st == 2)) and np.all(np.isfinite(f[st == 1]))
    for s in np.nonzero(st == 1)[0]:
        p = np.exp(theta[s])
        lml, _ = gp.fit_grad(X[:20], yc[:20], "matern52", p[0], p[1], p[2], 1e-10, False)
        assert -f[s] == pytest.approx(lml, rel=1e-9, abs=1e-9)


def test_hyper_fit_starts_in_threads_is_the_sequential_result(ta):
    """N > 128: the starts of the hyper-parameter fit run side by side (one thread + one handle on a
    private stream each).  Every start walks the iterates it walks alone, so the outcome equals the
    one-after-the-other run bit for bit -- likelihood, hyper-parameters and evaluation count"""
    import time
    X, y, _ = _synth(9, 400, 5, 1)
    res = {}
    for name, above in (("threads", 128), ("sequential", None), ("auto", "auto")):
        sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, np.ones(5), 1e-2), normalize_y=True,
                                                  random_state=0), training_iterations=3, param_continuity=False,
                                parallel_restarts_above=above)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            sur.construct_model(0, X, y)
            t0 = time.perf_counter()
            model, info = sur.construct_model(1, X, y)
            ms = (time.perf_counter() - t0) * 1e3
        res[name] = (model.get_log_likelihood(), model.get_hyper_params(), info["lml_evaluations"], ms)
    a, b = res["threads"], res["sequential"]
    assert a[0] == b[0] and a[2] == b[2]
    np.testing.assert_array_equal(a[1], b[1])
    c = res["auto"]                                   # the default: side by side at this size
    assert c[0] == b[0] and c[2] == b[2]
    np.testing.assert_array_equal(c[1], b[1])
    print("hyper-parameter fit, N=400, 3 starts, %d evaluations: %.1f ms in threads, %.1f ms one after the other" % (a[2], a[3], b[3]))


def test_private_stream_toggle(ta):
    """tgp_set_private_stream: the same fit + sweep on the shared stream, on a private one and back
    on the shared one give identical results; handles on private streams work from several threads"""
    import threading
    X, y, Xc = _synth(21, 300, 4, 2000)
    gp = ta.NativeGP(0, "f64")
    def run(g):
        lml, _, _ = g.fit(X, y, "matern52", 1.2, 0.7, 1e-3, 1e-10, True)
        g.set_candidates(Xc)
        r = g.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True)
        return lml, r["mu"].copy(), r["sigma"].copy(), r["best_idx"]
    a = run(gp)
    gp.set_private_stream(True)
    b = run(gp)
    gp.set_private_stream(True)                     # (idempotent)
    gp.set_private_stream(False)
    c = run(gp)
    for other in (b, c):
        assert other[0] == a[0] and other[3] == a[3]
        np.testing.assert_array_equal(other[1], a[1])
        np.testing.assert_array_equal(other[2], a[2])
    gps = [ta.NativeGP(0, "f64") for _ in range(3)]
    out = [None] * 3
    for g in gps:
        g.set_private_stream(True)
    def work(i):
        for _ in range(5):
            out[i] = run(gps[i])
    ths = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    for o in out:
        assert o[0] == a[0] and o[3] == a[3]
        np.testing.assert_array_equal(o[1], a[1])
    del gps


@pytest.mark.parametrize("kind,N,D,ard", [("matern52", 150, 3, False), ("rbf", 300, 5, True), ("matern32", 500, 8, False),
                                          ("matern52", 1000, 4, False)])
def test_device_optimizer_above_128(ta, kind, N, D, ard):
    """N > 128: optimizer='device' and the default both run L-BFGS-B inside the library (csrc/host_lbfgsb.hpp), a C++
    thread and a stream per start driving tgp_fit_grad -- no SciPy, no interpreter between two evaluations.  From the
    same starts they walk what SciPy walks when it drives the same GPU objective (optimizer='scipy', the default of
    rounds 1-4): the same optimum, and the same number of evaluations wherever the walk is short of the point where
    rounding decides (tests/test_host_lbfgs.py holds the optimiser to SciPy's exact walk on the CPU; on the log marginal
    likelihood a last-bit difference between the two implementations' small dense solves can grow to a different trial
    step within ten evaluations -- N = 150 below: 36 / 39 / 30 evaluations under SciPy, 49 / 39 / 34 here)."""
    X, y, _ = _synth(5 + N, N, D, 1)
    ls = np.full(D, 0.8) if ard else 0.8
    res = []
    for opt in ("scipy", "device", "fmin_l_bfgs_b"):
        sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel(kind, 1.0, ls, 1e-2), normalize_y=True,
                                                  random_state=0, optimizer=opt), training_iterations=3)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            model, info = sur.construct_model(0, X, y)
        res.append((model.get_log_likelihood(), model.get_hyper_params(), info["lml_evaluations"]))
        sur.close()
    (l_ref, th_ref, ev_ref), (l_dev, th_dev, ev_dev), (l_def, th_def, ev_def) = res
    assert (l_dev, ev_dev) == (l_def, ev_def) and np.array_equal(th_dev, th_def)
    assert abs(l_def - l_ref) <= 1e-8 * abs(l_ref), (l_ref, l_def)
    np.testing.assert_allclose(np.log(th_def), np.log(th_ref), atol=5e-3)     # (a flat optimum: the LML above is the sharp check)
    assert 0.6 * ev_ref - 10 <= ev_def <= 1.5 * ev_ref + 25, (ev_def, ev_ref)


@pytest.mark.parametrize("N", [400, 1300])        # Np = 512 (one outer block) | 1536 (the workers' inverse in line, batched over the blocks)
def test_device_optimizer_above_128_does_not_depend_on_the_thread_count(ta, N):
    """every start walks its own iterates on its own handle: one thread (the caller's handle, shared streams) or four
    (pooled workers on private streams), the same theta bit for bit"""
    import subprocess
    import sys
    child = ("import sys, numpy as np; sys.path.insert(0, %r); import turbo_amd as ta\n"
             "rng = np.random.RandomState(3); X = rng.uniform(0, 1, (NN, 4)); y = np.sin(3 * X.sum(1)) + 0.05 * rng.normal(size=NN)\n"
             "gp = ta.NativeGP(0, 'f64')\n"
             "th0 = np.log(np.array([[1.0, 0.5, 1e-2], [0.3, 2.0, 1e-3], [5.0, 0.1, 1e-1], [2.0, 1.0, 1e-4]]))\n"
             "b = np.log(np.array([[1e-5, 1e5]] * 3))\n"
             "th, f, st, ev = gp.fit_optimise(X, y, 'matern52', th0, 1, b, 1e-10, True)\n"
             "print(th.tobytes().hex(), f.tobytes().hex(), st.tolist(), ev)\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))).replace("NN", str(N))
    outs = []
    for threads in ("1", "4"):
        out = subprocess.run([sys.executable, "-c", child], env=dict(os.environ, TGP_HYPER_THREADS=threads),
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-3000:]
        outs.append(out.stdout)
    assert outs[0] == outs[1] and ("[1, 1, 1, 1]" in outs[0] or N != 400)


# ---- "next" row SURVEY 8(f)3: one-row incremental fit ---------------------------------------

@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_incremental_fit_matches_full_refit(ta, dtype):
    """the Optimiser appends one trial per iteration (turbo/optimiser.py:335-336): the appended
    factor must equal a from-scratch fit to rounding, at every step, including across a padding
    boundary (N = 256 -> 257 falls back to the full path)"""
    X, y, Xc = _synth(31, 262, 5, 300)
    kern = ("matern52", 1.3, 0.8, 2e-3)
    inc = ta.NativeGP(0, dtype)
    full = ta.NativeGP(0, dtype)
    n0 = 248
    inc.fit(X[:n0], y[:n0], *kern, 1e-10, True, append=True)
    assert not inc.appended
    for n in range(n0 + 1, 262):
        lml_i, ym_i, ys_i = inc.fit(X[:n], y[:n], *kern, 1e-10, True, append=True)
        lml_f, ym_f, ys_f = full.fit(X[:n], y[:n], *kern, 1e-10, True)
        assert inc.appended == (n != 257), n          # 257 needs a new 256-row padding block
        assert (ym_i, ys_i) == (ym_f, ys_f)
        assert lml_i == pytest.approx(lml_f, rel=1e-11, abs=1e-9)
        np.testing.assert_allclose(inc.debug_read(ta._lib.BUF_L), full.debug_read(ta._lib.BUF_L), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(inc.debug_read(ta._lib.BUF_ALPHA), full.debug_read(ta._lib.BUF_ALPHA), rtol=1e-6, atol=1e-7)
        inc.set_candidates(Xc)
        full.set_candidates(Xc)
        ri = inc.sweep(ta._lib.ACQ_EI, -1.0, float(y[:n].min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)
        rf = full.sweep(ta._lib.ACQ_EI, -1.0, float(y[:n].min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)
        tol = 1e-9 if dtype == "f64" else 1e-5
        np.testing.assert_allclose(ri["mu"], rf["mu"], rtol=tol, atol=tol)
        np.testing.assert_allclose(ri["sigma"] ** 2, rf["sigma"] ** 2, rtol=tol, atol=tol)
        # ... and, independently of the HIP full fit, the CPU oracle's fit of the same n rows
        om = o.fit(X[:n], y[:n], *kern, 1e-10, True)
        assert lml_i == pytest.approx(om.lml, rel=1e-9, abs=1e-9), n
        assert (ym_i, ys_i) == (pytest.approx(om.y_mean, rel=1e-13, abs=1e-14), pytest.approx(om.y_std, rel=1e-13))
        np.testing.assert_allclose(inc.debug_read(ta._lib.BUF_L), np.tril(om.L), rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(inc.debug_read(ta._lib.BUF_ALPHA), om.alpha.ravel(), rtol=1e-6,
                                   atol=1e-8 * np.abs(om.alpha).max())
        omu, osig = o.predict(om, Xc)
        oacq = o.acquisition("ei", omu, osig, "min", 0.01, float(y[:n].min()))
        if dtype == "f64":
            np.testing.assert_allclose(ri["mu"], omu, rtol=RTOL, atol=1e-9)
            np.testing.assert_allclose(ri["sigma"] ** 2, osig ** 2, rtol=RTOL, atol=VAR_ATOL * (kern[1] + kern[3]) * om.y_std ** 2)
            np.testing.assert_allclose(ri["acq"], oacq, rtol=RTOL, atol=1e-12)
        else:
            assert np.max(np.abs(ri["mu"] - omu)) < F32_MU_TOL * om.y_std
            assert np.max(np.abs(ri["sigma"] ** 2 - osig ** 2)) < F32_VAR_TOL * (kern[1] + kern[3]) * om.y_std ** 2
    # anything but "same prefix + one row, same hyper-parameters" refits from scratch
    inc.fit(X[:100], y[:100], *kern, 1e-10, True, append=True)
    inc.fit(X[:101], y[:101], kern[0], kern[1], 0.9, kern[3], 1e-10, True, append=True)
    assert not inc.appended                       # length scale changed
    Xm = X[:102].copy()
    Xm[3, 1] += 1e-9
    inc.fit(Xm, y[:102], kern[0], kern[1], 0.9, kern[3], 1e-10, True, append=True)
    assert not inc.appended                       # prefix mutated
    # an appended duplicate row with no noise and no jitter is singular: same error as the full path
    inc.fit(Xm, y[:102], kern[0], kern[1], 0.9, 0.0, 0.0, True, append=True)
    with pytest.raises(np.linalg.LinAlgError):
        inc.fit(np.vstack([Xm, Xm[:1]]), np.append(y[:102], y[0]), kern[0], kern[1], 0.9, 0.0, 0.0, True, append=True)
    inc.fit(Xm, y[:102], kern[0], kern[1], 0.9, 1e-3, 1e-10, True, append=True)   # still usable
    assert not inc.appended


def test_incremental_fit_into_a_skipped_panel(ta):
    """the full fit skips Cholesky panels that hold only padding (N = 60: panel 0 alone); rows
    appended afterwards land in those never-factored panels and must still match a fresh fit"""
    X, y, Xc = _synth(32, 140, 4, 200)
    kern = ("rbf", 1.0, 0.6, 1e-3)
    inc, full = ta.NativeGP(0, "f64"), ta.NativeGP(0, "f64")
    inc.fit(X[:60], y[:60], *kern, 1e-10, True, append=True)
    for n in range(61, 140):
        lml_i, _, _ = inc.fit(X[:n], y[:n], *kern, 1e-10, True, append=True)
        assert inc.appended
        if n in (61, 64, 65, 66, 100, 128, 129, 139):
            lml_f, _, _ = full.fit(X[:n], y[:n], *kern, 1e-10, True)
            assert lml_i == pytest.approx(lml_f, rel=1e-11, abs=1e-9)
            np.testing.assert_allclose(inc.debug_read(ta._lib.BUF_L), full.debug_read(ta._lib.BUF_L), rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(inc.debug_read(ta._lib.BUF_LINV), full.debug_read(ta._lib.BUF_LINV), rtol=1e-7, atol=1e-9)
            inc.set_candidates(Xc); full.set_candidates(Xc)
            ri = inc.sweep(ta._lib.ACQ_EI, -1.0, float(y[:n].min()), 0.01, want_mu=True, want_sigma=True)
            rf = full.sweep(ta._lib.ACQ_EI, -1.0, float(y[:n].min()), 0.01, want_mu=True, want_sigma=True)
            np.testing.assert_allclose(ri["mu"], rf["mu"], rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(ri["sigma"] ** 2, rf["sigma"] ** 2, rtol=1e-8, atol=1e-10)
            om = o.fit(X[:n], y[:n], *kern, 1e-10, True)       # the oracle's fit of the same rows
            assert lml_i == pytest.approx(om.lml, rel=1e-9, abs=1e-9)
            np.testing.assert_allclose(inc.debug_read(ta._lib.BUF_L), np.tril(om.L), rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(inc.debug_read(ta._lib.BUF_ALPHA), om.alpha.ravel(), rtol=1e-6,
                                       atol=1e-8 * np.abs(om.alpha).max())
            omu, osig = o.predict(om, Xc)
            np.testing.assert_allclose(ri["mu"], omu, rtol=RTOL, atol=1e-9)
            np.testing.assert_allclose(ri["sigma"] ** 2, osig ** 2, rtol=RTOL, atol=VAR_ATOL * (kern[1] + kern[3]) * om.y_std ** 2)
            # the gradient path refits in full from the appended state's inputs
            g_i = inc.acq_grad(Xc[:4], ta._lib.ACQ_EI, -1.0, float(y[:n].min()), 0.01)
            g_f = full.acq_grad(Xc[:4], ta._lib.ACQ_EI, -1.0, float(y[:n].min()), 0.01)
            np.testing.assert_allclose(g_i[1], g_f[1], rtol=1e-6, atol=1e-9)


def test_incremental_fit_through_the_plugin(ta):
    X, y, Xc = _synth(41, 60, 3, 200)
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("rbf", 1.0, 0.7, 1e-3), optimizer=None,
                                              normalize_y=True), training_iterations=1)
    ref = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("rbf", 1.0, 0.7, 1e-3), optimizer=None,
                                              normalize_y=True), training_iterations=1, incremental=False)
    for n in range(50, 60):
        m, _ = sur.construct_model(n, X[:n], y[:n])
        r, _ = ref.construct_model(n, X[:n], y[:n])
        assert m.appended == (n > 50) and not r.appended
        np.testing.assert_allclose(m.predict(Xc), r.predict(Xc), rtol=1e-9, atol=1e-10)
        assert m.get_log_likelihood() == pytest.approx(r.get_log_likelihood(), rel=1e-11)
        om = o.fit(X[:n], y[:n], "rbf", 1.0, 0.7, 1e-3, 1e-10, True)
        assert m.get_log_likelihood() == pytest.approx(om.lml, rel=1e-9)
        mu, sg = m.predict(Xc, return_std_dev=True)
        omu, osig = o.predict(om, Xc)
        np.testing.assert_allclose(mu, omu, rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(sg ** 2, osig ** 2, rtol=RTOL, atol=VAR_ATOL * (1 + 1e-3) * om.y_std ** 2)


# ---- "next" row SURVEY 8(f)4: candidates drawn on the device ----------------------------------

def test_device_candidate_generator(ta):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from philox_ref import uniform_candidates
    X, y, _ = _synth(3, 64, 5, 1)
    gp = ta.NativeGP(0, "f64")
    gp.fit(X, y, "rbf", 1.0, 0.9, 1e-3, 1e-10, True)
    lo, hi = np.array([-5.0, 0.0, 1.0, -1.0, 10.0]), np.array([10.0, 15.0, 2.0, 1.0, 11.0])
    M = 5001
    gp.gen_candidates(1234567890123, 0, M, lo, hi)
    got = np.vstack([gp.get_candidate(i) for i in (0, 1, 2, 777, M - 1)])
    want = uniform_candidates(1234567890123, 0, M, lo, hi)
    np.testing.assert_array_equal(got, want[[0, 1, 2, 777, M - 1]])       # bit-exact stream
    assert np.all(want >= lo) and np.all(want < hi)
    # the sweep over the generated batch equals the sweep over the same batch uploaded
    r1 = gp.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_acq=True)
    gp.set_candidates(want)
    r2 = gp.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_acq=True)
    np.testing.assert_array_equal(r1["acq"], r2["acq"])
    assert r1["best_idx"] == r2["best_idx"]
    # shards are disjoint pieces of one stream
    gp.gen_candidates(1234567890123, 3000, 100, lo, hi)
    np.testing.assert_array_equal(gp.get_candidate(5), want[3005])
    # through the plugin: deterministic in the seed, inside the bounds, a different batch per call
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("rbf", 1.0, 0.9, 1e-3), optimizer=None,
                                              normalize_y=True), training_iterations=1)
    model, _ = sur.construct_model(0, X, y)
    f, _ = ta.EI(0.01).construct_function(0, model, "min", float(y.min()))
    b = ta.Bounds([("p%d" % i, lo[i], hi[i]) for i in range(5)])
    x1, i1 = ta.CandidateSweep(num_random=M, device_rng_seed=99)(b, f)
    x2, i2 = ta.CandidateSweep(num_random=M, device_rng_seed=99)(b, f)
    np.testing.assert_array_equal(x1, x2)
    assert i1["max_acq"] == i2["max_acq"] and np.all(x1 >= lo) and np.all(x1 <= hi)
    assert i1["sweep_ms"] > 0          # device time of the sweep (hipEvents), reported beside max_acq
    c99 = uniform_candidates(99, 0, M, lo, hi)
    vals = f(c99)
    assert i1["max_acq"] == vals.max() and np.array_equal(x1[0], c99[int(np.argmax(vals))])


def test_device_lhs_design(ta):
    """tgp_lhs_design / tgp_gen_candidates_lhs against the NumPy restatement (tests/philox_ref.py),
    bit for bit; a Latin hypercube (every stratum of every dimension hit exactly once); shards are
    rows of the one design; the selector mirrors LHS_selector's contract"""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from philox_ref import lhs_design
    lo, hi = np.array([-5.0, 0.0, 1.0, -1.0, 10.0]), np.array([10.0, 15.0, 2.0, 1.0, 11.0])
    gp = ta.NativeGP(0, "f64")
    for n in (1, 7, 64, 1000, 4097):
        got = gp.lhs_design(987654321012, 0, n, n, lo, hi)                 # no model needed
        want = lhs_design(987654321012, 0, n, n, lo, hi)
        np.testing.assert_array_equal(got, want)
        strata = np.floor((got - lo) / (hi - lo) * n).astype(int)
        for d in range(5):
            assert sorted(strata[:, d].tolist()) == list(range(n))
        assert np.all(got >= lo) and np.all(got < hi)
    whole = lhs_design(5, 0, 1000, 1000, lo, hi)
    np.testing.assert_array_equal(gp.lhs_design(5, 300, 250, 1000, lo, hi), whole[300:550])   # a shard
    with pytest.raises(ValueError):
        gp.lhs_design(5, 900, 200, 1000, lo, hi)                            # sequence exhausted
    # as the resident candidate batch of a sweep
    X, y, _ = _synth(3, 64, 5, 1)
    gp.fit(X, y, "rbf", 1.0, 0.9, 1e-3, 1e-10, True)
    gp.gen_candidates_lhs(5, 100, 700, 1000, lo, hi)
    np.testing.assert_array_equal(gp.read_candidates(), whole[100:800])
    r1 = gp.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_acq=True)
    r2 = gp.evaluate(whole[100:800], ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_acq=True)
    np.testing.assert_array_equal(r1["acq"], r2["acq"])
    # the selector: consecutive slices of one design, like the reference's LHS_selector
    b = ta.Bounds([("p%d" % i, lo[i], hi[i]) for i in range(5)])
    sel = ta.LHS_selector(num_total=10, device_seed=77)
    first, second = sel(4, b), sel(6, b)
    np.testing.assert_array_equal(np.vstack([first, second]), lhs_design(77, 0, 10, 10, lo, hi))
    with pytest.raises(AssertionError):
        sel(1, b)
    np.random.seed(0)
    host = ta.LHS_selector(num_total=8)(8, b)                               # host design (the reference's draw order)
    strata = np.floor((host - lo) / (hi - lo) * 8).astype(int)
    assert all(sorted(strata[:, d].tolist()) == list(range(8)) for d in range(5))
    # through the sweep: the whole candidate batch is one design
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("rbf", 1.0, 0.9, 1e-3), optimizer=None,
                                              normalize_y=True), training_iterations=1)
    model, _ = sur.construct_model(0, X, y)
    f, _ = ta.EI(0.01).construct_function(0, model, "min", float(y.min()))
    x1, i1 = ta.CandidateSweep(num_random=3000, device_rng_seed=99, device_design='lhs')(b, f)
    c99 = lhs_design(99, 0, 3000, 3000, lo, hi)
    vals = f(c99)
    assert i1["max_acq"] == vals.max() and np.array_equal(x1[0], c99[int(np.argmax(vals))])


# ---- "next" row SURVEY 8(f)2: acquisition gradients and the gradient stage ------------------------

@pytest.mark.parametrize("kind,ard", [("rbf", False), ("matern52", True), ("matern32", False), ("matern12", False)])
def test_acquisition_gradient_vs_finite_differences(ta, kind, ard):
    """closed-form d acq / d x from the GPU against central differences of the oracle"""
    X, y, _ = _synth(51, 120, 4, 1)
    ls = np.array([0.5, 0.8, 1.1, 1.4]) if ard else 0.9
    c, noise = 1.6, 5e-3
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel(kind, c, ls, noise), optimizer=None,
                                              normalize_y=True), training_iterations=1)
    model, _ = sur.construct_model(0, X, y)
    om = o.fit(X, y, kind, c, ls, noise, 1e-10, True)
    rng = np.random.RandomState(5)
    Xq = rng.uniform(0.05, 0.95, size=(7, 4))
    inc = float(y.min())
    for cls, param, okind in ((ta.EI, 0.01, "ei"), (ta.PI, 0.01, "pi"), (ta.UCB, 2.0, "ucb"), (ta.UCB, float("inf"), "ucb")):
        for ext in ("min", "max"):
            fac = cls(param)
            args = [0, model, ext] + ([inc] if fac.get_type() == "improvement" else [])
            f, _ = fac.construct_function(*args)
            val, grad = f.value_and_grad(Xq)

            def oracle_val(P):
                mu, sg = o.predict(om, P)
                return o.acquisition(okind, mu, sg, ext, param, inc)
            np.testing.assert_allclose(val, oracle_val(Xq), rtol=1e-7, atol=1e-12)
            np.testing.assert_allclose(val, f(Xq), rtol=1e-9, atol=1e-13)     # same as the sweep path
            h = 1e-6
            fd = np.empty_like(grad)
            for d in range(4):
                e = np.zeros(4); e[d] = h
                fd[:, d] = (oracle_val(Xq + e) - oracle_val(Xq - e)) / (2 * h)
            scale = np.abs(fd).max() + 1e-12
            # central differences of an O(1) function at h = 1e-6 carry ~1e-10 of rounding noise
            np.testing.assert_allclose(grad, fd, rtol=2e-4, atol=max(2e-6 * scale, 2e-9))


def test_gradient_stage_vs_reference(ta):
    """RandomAndQuasiNewton with grad_restarts > 0 (auxiliary_optimisers.py:69-112) replayed on the
    batches the reference drew: the refined point must be at least as good as the reference's
    (which used finite-difference gradients) and land on the same optimum"""
    with np.load(golden_path("stage2_branin"), allow_pickle=False) as z:
        t = {k: z[k] for k in z.files}
    X, y = t["X"], t["y"]
    b = ta.Bounds([("x", -5.0, 10.0), ("y", 0.0, 15.0)])
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 2.0, 3.0, 1e-2), optimizer=None,
                                              normalize_y=True), training_iterations=1)
    model, _ = sur.construct_model(0, X, y)
    for name, fac, args in (("ei", ta.EI(xi=0.01), [float(y.min())]), ("ucb", ta.UCB(beta=2.0), [])):
        f, _ = fac.construct_function(0, model, "min", *args)
        batches = [t[name + "_batch"], t[name + "_starts"]]
        aux = ta.RandomAndQuasiNewton(num_random=256, grad_restarts=6, start_from_best=2,
                                      gen_random=lambda n, lb, it=iter(batches): next(it))
        x, info = aux(b, f)
        want = float(t[name + "_max_acq"])
        assert float(np.max(f(t[name + "_batch"]))) == pytest.approx(float(t[name + "_random_best"]), rel=1e-7)
        assert info["max_acq"] >= want - 1e-6 * abs(want)          # at least as good
        assert info["max_acq"] == pytest.approx(want, rel=1e-4)    # same optimum
        np.testing.assert_allclose(x, t[name + "_x"], atol=5e-3)
        assert info["max_acq"] > float(t[name + "_random_best"])   # the stage did improve on the sweep


def test_gradient_stage_lockstep_equals_sequential(ta):
    """the restarts advance in lock-step over batched tgp_acq_grad calls: a point's value and
    gradient do not depend on what else is in the batch, so the outcome is the sequential one"""
    import time
    X, y, _ = _synth(17, 900, 6, 1)
    b = ta.Bounds([("x%d" % d, 0.0, 1.0) for d in range(6)])
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 0.8, 1e-3), optimizer=None,
                                              normalize_y=True), training_iterations=1)
    model, _ = sur.construct_model(0, X, y)
    f, _ = ta.EI(xi=0.01).construct_function(0, model, "min", float(y.min()))
    # batch invariance of the native gradient call itself, bitwise
    P = np.random.RandomState(5).uniform(0, 1, (10, 6))
    v_all, g_all = f.value_and_grad(P)
    for i in (0, 3, 9):
        v1, g1 = f.value_and_grad(P[i:i + 1])
        assert v1[0] == v_all[i] and np.array_equal(g1[0], g_all[i])
    out = {}
    for mode in ("scipy", False, True):        # SciPy per restart at a rendezvous | SciPy one restart after the other | L-BFGS-B in the library
        np.random.seed(23)
        aux = ta.RandomAndQuasiNewton(num_random=2000, grad_restarts=10, start_from_best=2, lockstep=mode)
        t0 = time.perf_counter()
        x, info = aux(b, f)
        out[mode] = (x, info["max_acq"], time.perf_counter() - t0, aux.last_batches, info)
    np.testing.assert_array_equal(out["scipy"][0], out[False][0])
    assert out["scipy"][1] == out[False][1]
    assert out["scipy"][3][0] == 10 and len(out["scipy"][3]) < sum(out["scipy"][3]) / 2   # batched: far fewer native calls than points evaluated
    # the library's own L-BFGS-B (tgp_acq_lbfgsb, the default): the same point to the optimiser's rounding
    np.testing.assert_allclose(out[True][0], out[False][0], atol=1e-9)
    assert abs(out[True][1] - out[False][1]) <= 1e-12 * max(1.0, abs(out[False][1]))
    assert 0.7 * sum(out["scipy"][3]) <= out[True][4]["gradient_evaluations"] <= 1.4 * sum(out["scipy"][3])
    print("gradient stage: library %.1f ms, SciPy lock-step %.1f ms in %d batched calls, sequential %.1f ms in %d calls"
          % (out[True][2] * 1e3, out["scipy"][2] * 1e3, len(out["scipy"][3]), out[False][2] * 1e3, sum(out["scipy"][3])))


def test_sweep_topk_matches_argsort(ta):
    """tgp_sweep_topk against argsort of the (M,) acquisition vector: several reduction levels,
    ties (duplicated candidates) broken by the lowest index, k = 1, fewer candidates than k"""
    X, y, _ = _synth(71, 300, 5, 1)
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 0.8, 1e-3), optimizer=None,
                                              normalize_y=True), training_iterations=1)
    model, _ = sur.construct_model(0, X, y)
    f, _ = ta.EI(0.01).construct_function(0, model, "min", float(y.min()))
    rng = np.random.RandomState(3)
    for M in (10, 5000, 20000, 310000):
        Xc = rng.uniform(0, 1, (M, 5))
        if M > 100:
            full0 = f(Xc)
            b = int(np.argmax(full0))
            Xc[3] = Xc[b]                      # the best candidate three times: ties
            Xc[M - 2] = Xc[b]
        full = f(Xc)
        order = np.lexsort((np.arange(M), -full))
        for k in (1, 7, 64):
            idx, vals = f.maximise_topk(Xc, k)
            kk = min(k, M)
            assert idx.shape == vals.shape == (kk,)
            np.testing.assert_array_equal(idx, order[:kk])
            np.testing.assert_array_equal(vals, full[order[:kk]])
        assert int(idx[0]) == f.maximise(Xc)[0]
    lib = ta._lib.load()
    import ctypes
    ctx = sur._context()
    v = np.empty(65); i = np.empty(65, dtype=np.int64)
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)
    assert lib.tgp_sweep_topk(ctx._h, 3, -1.0, 0.0, 0.01, 65, v.ctypes.data_as(dp), i.ctypes.data_as(ip), None) == ta._lib.BAD_ARG
    assert lib.tgp_sweep_topk(ctx._h, 0, -1.0, 0.0, 0.01, 5, v.ctypes.data_as(dp), i.ctypes.data_as(ip), None) == ta._lib.BAD_ARG


def test_gradient_stage_on_device_vs_reference(ta):
    """the same reference fixture as test_gradient_stage_vs_reference, through the ON-DEVICE
    optimiser (tgp_sweep_topk for the best random starts, tgp_acq_refine for the restarts)"""
    with np.load(golden_path("stage2_branin"), allow_pickle=False) as z:
        t = {k: z[k] for k in z.files}
    X, y = t["X"], t["y"]
    b = ta.Bounds([("x", -5.0, 10.0), ("y", 0.0, 15.0)])
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 2.0, 3.0, 1e-2), optimizer=None,
                                              normalize_y=True), training_iterations=1)
    model, _ = sur.construct_model(0, X, y)
    for name, fac, args in (("ei", ta.EI(xi=0.01), [float(y.min())]), ("ucb", ta.UCB(beta=2.0), [])):
        f, _ = fac.construct_function(0, model, "min", *args)
        batches = [t[name + "_batch"], t[name + "_starts"]]
        aux = ta.RandomAndQuasiNewton(num_random=256, grad_restarts=6, start_from_best=2, on_device=True,
                                      gen_random=lambda n, lb, it=iter(batches): next(it))
        x, info = aux(b, f)
        want = float(t[name + "_max_acq"])
        assert info["max_acq"] >= want - 1e-6 * abs(want)          # at least as good as the reference's
        assert info["max_acq"] == pytest.approx(want, rel=1e-4)    # same optimum
        np.testing.assert_allclose(x, t[name + "_x"], atol=5e-3)
        assert info["max_acq"] > float(t[name + "_random_best"])
        assert info["refine_iterations"] > 0
        assert float(f(x)[0]) == pytest.approx(info["max_acq"], rel=1e-9)   # the reported value is the acquisition there


def test_on_device_optimiser_vs_scipy_lockstep(ta):
    """10 restarts at N = 900, 6D: the resident projected L-BFGS must reach what SciPy's L-BFGS-B
    reaches from the same starts (value within 1e-6 relative or better), inside the bounds; prints
    both latencies"""
    import time
    X, y, _ = _synth(17, 900, 6, 1)
    b = ta.Bounds([("x%d" % d, 0.0, 1.0) for d in range(6)])
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 0.8, 1e-3), optimizer=None,
                                              normalize_y=True), training_iterations=1)
    model, _ = sur.construct_model(0, X, y)
    out = {}
    for acq_name, fac, args in (("ei", ta.EI(xi=0.01), [float(y.min())]), ("ucb", ta.UCB(beta=2.0), [])):
        f, _ = fac.construct_function(0, model, "min", *args)
        for mode in ("device", "scipy"):
            np.random.seed(23)
            aux = ta.RandomAndQuasiNewton(num_random=2000, grad_restarts=10, start_from_best=2,
                                          on_device=(mode == "device"))
            aux(b, f)                                  # warm
            np.random.seed(23)
            t0 = time.perf_counter()
            x, info = aux(b, f)
            out[mode] = (x, info, (time.perf_counter() - t0) * 1e3)
        xd, infod, msd = out["device"]
        xs, infos, mss = out["scipy"]
        assert np.all(xd >= 0.0) and np.all(xd <= 1.0)
        assert infod["max_acq"] >= infos["max_acq"] - 1e-6 * abs(infos["max_acq"]), (acq_name, infod, infos)
        print("gradient stage %s, 10 restarts, N=900: on device %.2f ms (%d evaluations), SciPy lock-step %.2f ms; "
              "max_acq %.9g vs %.9g" % (acq_name, msd, infod["refine_iterations"], mss, infod["max_acq"], infos["max_acq"]))
    # the raw entry point: a start on the boundary with the gradient pointing outward stays put
    f, _ = ta.UCB(beta=0.0).construct_function(0, model, "max")
    ctx = sur._context()
    P = np.random.RandomState(4).uniform(0, 1, (5, 6))
    xr, vr, st, its = ctx.acq_refine(P, np.zeros(6), np.ones(6), *f._native_args()[:1], 1.0, 0.0, 0.0, 300)
    assert np.all(st >= 1) and np.all(xr >= 0) and np.all(xr <= 1)
    v0, _ = f.value_and_grad(P)
    v1, g1 = f.value_and_grad(xr)
    assert np.all(v1 >= v0 - 1e-12)                    # never worse than the start
    np.testing.assert_allclose(vr, v1, rtol=1e-12, atol=1e-12)
    inner = (xr > 1e-9) & (xr < 1 - 1e-9)
    assert np.max(np.abs(g1[inner])) < 1e-3            # stationary in the free coordinates


@pytest.mark.parametrize("kind,N,D,ard", [("matern52", 30, 2, False), ("rbf", 64, 5, True), ("matern32", 65, 3, False),
                                          ("matern52", 100, 16, True), ("matern12", 128, 8, False),
                                          ("matern52", 128, 64, False)])
def test_one_launch_optimiser_small_problems(ta, kind, N, D, ard):
    """N <= 128: tgp_acq_refine runs every restart to the end inside ONE launch (small_refine_kernel).
    From the same starts SciPy's L-BFGS-B on the library's own value + gradient must not find a
    better optimum; every restart ends inside the bounds, never below its start, at a point that
    is stationary in the free coordinates, and reports the acquisition's value there"""
    from scipy.optimize import minimize
    X, y, _ = _synth(300 + N + D, N, D, 1)
    ls = (0.4 + 0.05 * np.arange(D)) if ard else 0.3 * np.sqrt(D)
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel(kind, 1.3, ls, 1e-3), optimizer=None,
                                              normalize_y=True), training_iterations=1)
    model, _ = sur.construct_model(0, X, y)
    ctx = sur._context()
    lo, hi = np.zeros(D), np.ones(D)
    P = np.random.RandomState(N).uniform(0, 1, (12, D))
    P[0] = X[int(np.argmin(y))]                                      # a start ON a training point
    P[1] = 0.0                                                       # a start in a corner
    for fac, args in ((ta.EI(xi=0.01), [float(y.min())]), (ta.UCB(beta=2.0), []), (ta.PI(xi=0.01), [float(y.min())])):
        f, _ = fac.construct_function(0, model, "min", *args)
        acq, inc, par = f._native_args()
        xr, vr, st, evals = ctx.acq_refine(P, lo, hi, acq, f.scale_factor, inc, par, 300)
        assert evals >= 1
        smooth = kind != "matern12"      # (the exponential kernel has a kink at every training point: a restart may end there by max_iter)
        assert np.all(st >= 1) or not smooth, (st, evals)
        assert np.all(xr >= lo) and np.all(xr <= hi)
        v0, _ = f.value_and_grad(P)
        v1, g1 = f.value_and_grad(xr)
        np.testing.assert_allclose(vr, v1, rtol=1e-9, atol=1e-12)
        assert np.all(v1 >= v0 - 1e-12)
        inner = (xr > 1e-9) & (xr < 1 - 1e-9)
        scale = max(1.0, float(np.max(np.abs(v1))))
        assert np.max(np.abs(g1[inner & (st >= 1)[:, None]]), initial=0.0) < 2e-3 * scale or not smooth
        best_scipy = -np.inf
        for r in range(len(P)):
            fun = lambda x: tuple((-v[0], -g[0]) for v, g in [f.value_and_grad(x[None, :])])[0]
            res = minimize(fun, np.clip(P[r], lo, hi), jac=True, method="L-BFGS-B", bounds=[(0.0, 1.0)] * D)
            best_scipy = max(best_scipy, -float(res.fun))
        assert float(vr.max()) >= best_scipy - 1e-6 * max(1.0, abs(best_scipy)), (float(vr.max()), best_scipy)


@pytest.mark.parametrize("N,D,R", [(1, 1, 1), (2, 1, 3), (3, 2, 200), (64, 1, 5), (65, 7, 5), (128, 63, 3), (128, 64, 2), (17, 5, 4096)])
def test_one_launch_optimiser_edge_shapes(ta, N, D, R):
    """the one-launch gradient stage at the edges of its domain: one training point, one dimension,
    D = 64, 4096 restarts, and a dimension whose bounds coincide -- every restart ends finite, inside
    the bounds and not below its start"""
    rng = np.random.RandomState(N + D + R)
    X = rng.uniform(0, 1, (N, D))
    y = np.sin(3 * X.sum(1)) + 0.1 * rng.normal(size=N)
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 0.5, 1e-2), optimizer=None,
                                              normalize_y=True), training_iterations=1)
    model, _ = sur.construct_model(0, X, y)
    ctx = sur._context()
    for fac, args in ((ta.EI(xi=0.01), [float(y.min())]), (ta.UCB(beta=2.0), [])):
        f, _ = fac.construct_function(0, model, "min", *args)
        acq, inc, par = f._native_args()
        P = rng.uniform(0, 1, (R, D))
        lo, hi = np.zeros(D), np.ones(D)
        if D > 1:
            lo[0] = hi[0] = 0.3
            P[:, 0] = 0.3
        xr, vr, st, ev = ctx.acq_refine(P, lo, hi, acq, f.scale_factor, inc, par, 200)
        assert ev >= 1 and np.all(np.isfinite(vr)) and np.all(xr >= lo) and np.all(xr <= hi)
        v0, _ = f.value_and_grad(P[:64])
        assert np.all(vr[:64] >= v0 - 1e-12)


@pytest.mark.parametrize("N,D", [(300, 65), (200, 100), (150, 256), (200, 257), (160, 700), (140, 1024), (130, 1025),
                                 (100, 2500), (90, 4096)])
def test_on_device_optimiser_above_64_dimensions(ta, N, D):
    """64 < D <= 4096 (the library's limit): the wave step with four / sixteen coordinates per lane
    (refine_step_wave_kernel<4>, <16>) and, beyond D = 1024, a team of eight waves per restart (<8, 8>); from the
    same starts SciPy's L-BFGS-B on the library's own value + gradient must not find a better optimum,
    and every restart converges inside the bounds to the value the acquisition has there"""
    from scipy.optimize import minimize
    X, y, _ = _synth(77 + D, N, D, 1)
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 0.3 * np.sqrt(D), 1e-3), optimizer=None,
                                              normalize_y=True), training_iterations=1)
    model, _ = sur.construct_model(0, X, y)
    ctx = sur._context()
    lo, hi = np.zeros(D), np.ones(D)
    P = np.random.RandomState(D).uniform(0, 1, (6, D))
    f, _ = ta.EI(xi=0.01).construct_function(0, model, "min", float(y.min()))
    acq, inc, par = f._native_args()
    xr, vr, st, ev = ctx.acq_refine(P, lo, hi, acq, f.scale_factor, inc, par, 400)
    assert np.all(st == 1), (st, ev)
    assert np.all(xr >= lo) and np.all(xr <= hi)
    v0, _ = f.value_and_grad(P)
    v1, _ = f.value_and_grad(xr)
    np.testing.assert_allclose(vr, v1, rtol=1e-9, atol=1e-12)
    assert np.all(v1 >= v0 - 1e-12)
    best_scipy = -np.inf
    for r in range(len(P)):
        fun = lambda x: tuple((-v[0], -g[0]) for v, g in [f.value_and_grad(x[None, :])])[0]
        res = minimize(fun, P[r], jac=True, method="L-BFGS-B", bounds=[(0.0, 1.0)] * D)
        best_scipy = max(best_scipy, -float(res.fun))
    assert float(vr.max()) >= best_scipy - 1e-6 * max(1.0, abs(best_scipy)), (float(vr.max()), best_scipy)


def test_predict_many_stored_models(ta):
    """the plot path: T stored models (one per trial, growing N, their own hyper-parameters) x one
    grid, as one library call per size class -- rows equal the per-model predict bit for bit (N <= 128; to
    rounding for 128 < N <= 256, whose batched fit factors its blocks in another order) and the oracle to
    the fp64 tolerance"""
    rng = np.random.RandomState(8)
    D, M = 2, 10000
    Xall = rng.uniform(0, 1, (200, D))
    yall = np.sin(5 * Xall[:, 0]) * np.cos(3 * Xall[:, 1]) + 0.01 * rng.normal(size=200)
    grid = rng.uniform(0, 1, (M, D))
    sizes = list(range(4, 40, 3)) + [63, 64, 65, 100, 127, 128, 150]
    sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 0.5, 1e-3), optimizer=None,
                                              normalize_y=True), training_iterations=1)
    models = []
    for k, n in enumerate(sizes):
        sur.model_params["kernel"] = ta.GPKernel("matern52", 1.0 + 0.1 * k, 0.3 + 0.02 * k if k % 2 else np.array([0.4, 0.6 + 0.01 * k]), 1e-3 * (1 + k))
        models.append(sur.construct_model(k, Xall[:n], yall[:n])[0])
    import time
    t0 = time.perf_counter()
    mus, sig = sur.predict_many(models, grid, return_std_dev=True)
    t_batch = time.perf_counter() - t0
    assert mus.shape == sig.shape == (len(sizes), M)
    t0 = time.perf_counter()
    single = [m.predict(grid, return_std_dev=True) for m in models]
    t_loop = time.perf_counter() - t0
    for t, (m1, s1) in enumerate(single):
        if sizes[t] <= 128:
            np.testing.assert_array_equal(mus[t], m1)
            np.testing.assert_array_equal(sig[t], s1)
        else:
            np.testing.assert_allclose(mus[t], m1, rtol=1e-9, atol=1e-10)
            np.testing.assert_allclose(sig[t] ** 2, s1 ** 2, rtol=1e-7, atol=1e-11)
    for t in (0, 5, len(sizes) - 2):
        m = models[t]
        om = o.fit(m.X, m.y, "matern52", m.kernel.constant, m.kernel.length_scale, m.kernel.noise_level, 1e-10, True)
        omu, osig = o.predict(om, grid)
        np.testing.assert_allclose(mus[t], omu, rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(sig[t] ** 2, osig ** 2, rtol=RTOL, atol=VAR_ATOL * (m.kernel.constant + m.kernel.noise_level) * om.y_std ** 2)
        assert m.get_log_likelihood() == pytest.approx(om.lml, rel=1e-9)
    np.testing.assert_array_equal(sur.predict_many(models[:3], grid[:7]), np.vstack([m.predict(grid[:7]) for m in models[:3]]))
    print("predict_many: %d models x %d points in %.2f ms, one by one %.2f ms" % (len(sizes), M, t_batch * 1e3, t_loop * 1e3))


def test_multi_device_handle_one_process(ta):
    """tgp_multi_*: several contexts in one process (here four on device 0), replicated fit,
    contiguous shards, host-side reduce -- identical to the single handle, for uploaded and for
    device-drawn candidates, with a tie across shards going to the lowest global index"""
    import ctypes
    lib = ta._lib.load()
    X, y, Xc = _synth(61, 300, 5, 50001)
    Xc[40000] = Xc[7]; Xc[20000] = Xc[7]            # the same row in three shards
    one = ta.NativeGP(0, "f64")
    lml1, _, _ = one.fit(X, y, "matern52", 1.0, 0.8, 1e-3, 1e-10, True)
    r1 = one.evaluate(Xc, ta._lib.ACQ_UCB, 1.0, 0.0, 2.0, want_acq=True)
    dp, ip = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int64)
    p = lambda a: a.ctypes.data_as(dp)
    m = ctypes.c_void_p()
    ids = (ctypes.c_int * 4)(0, 0, 0, 0)
    assert lib.tgp_multi_create(4, ids, 0, ctypes.byref(m)) == ta._lib.OK and lib.tgp_multi_size(m) == 4
    ls = np.array([0.8])
    lml = ctypes.c_double()
    assert lib.tgp_multi_fit(m, p(X), 300, 5, p(y), 3, 1.0, p(ls), 1, 1e-3, 1e-10, 1, ctypes.byref(lml), None, None) == ta._lib.OK
    assert lml.value == lml1
    assert lib.tgp_multi_set_candidates(m, p(Xc), Xc.shape[0]) == ta._lib.OK
    bv, bi = ctypes.c_double(), ctypes.c_int64()
    row, acq = np.empty(5), np.empty(Xc.shape[0])
    assert lib.tgp_multi_sweep(m, 1, 1.0, 0.0, 2.0, ctypes.byref(bv), ctypes.byref(bi), p(row), p(acq)) == ta._lib.OK
    np.testing.assert_array_equal(acq, r1["acq"])
    assert (bv.value, bi.value) == (r1["best_val"], r1["best_idx"]) and np.array_equal(row, Xc[bi.value])
    # a three-way tie on the winner: make the tied row the best one
    Xt = Xc.copy(); b = r1["best_idx"]; Xt[7] = Xc[b]; Xt[20000] = Xc[b]; Xt[40000] = Xc[b]
    assert lib.tgp_multi_set_candidates(m, p(Xt), Xt.shape[0]) == ta._lib.OK
    assert lib.tgp_multi_sweep(m, 1, 1.0, 0.0, 2.0, ctypes.byref(bv), ctypes.byref(bi), p(row), None) == ta._lib.OK
    assert bi.value == min(7, b) and bv.value == r1["best_val"]
    # device-drawn candidates: the shards are rows of ONE stream
    lo, hi = np.zeros(5), np.ones(5)
    assert lib.tgp_multi_gen_candidates(m, 2024, 30001, p(lo), p(hi)) == ta._lib.OK
    assert lib.tgp_multi_sweep(m, 3, -1.0, float(y.min()), 0.01, ctypes.byref(bv), ctypes.byref(bi), p(row), None) == ta._lib.OK
    one.gen_candidates(2024, 0, 30001, lo, hi)
    r2 = one.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01)
    assert (bv.value, bi.value) == (r2["best_val"], r2["best_idx"]) and np.array_equal(row, one.get_candidate(bi.value))
    assert lib.tgp_multi_create(2, (ctypes.c_int * 2)(0, 99), 0, ctypes.byref(ctypes.c_void_p())) == ta._lib.BAD_ARG
    assert b"device" in lib.tgp_multi_last_error(None)
    assert lib.tgp_multi_destroy(m) == ta._lib.OK


def test_c_abi_error_codes(ta):
    """status codes at the C boundary (include/turbogp.h): raw ctypes calls, no Python checks"""
    import ctypes
    lib = ta._lib.load()
    h = ctypes.c_void_p()
    assert lib.tgp_create(0, 7, ctypes.byref(h)) == ta._lib.BAD_ARG            # unknown dtype
    assert lib.tgp_create(999, 0, ctypes.byref(h)) == ta._lib.BAD_ARG          # no such device
    assert b"device" in lib.tgp_last_error(None)
    assert lib.tgp_create(0, 0, ctypes.byref(h)) == ta._lib.OK
    dp = ctypes.POINTER(ctypes.c_double)
    X = np.random.RandomState(0).rand(6, 2)
    y = np.arange(6.0)
    ls = np.array([0.5])
    p = lambda a: a.ctypes.data_as(dp)
    best = ctypes.c_double()
    assert lib.tgp_sweep(h, 3, -1.0, 0.0, 0.01, None, None, None, ctypes.byref(best), None, None) == ta._lib.NOT_FITTED
    assert lib.tgp_set_candidates(h, p(X), 6) == ta._lib.NOT_FITTED
    args = lambda kern=0, c=1.0, nls=1, noise=1e-3: (h, p(X), 6, 2, p(y), kern, c, p(ls), nls, noise, 1e-10, 1, None, None, None)
    assert lib.tgp_fit(*args(kern=9)) == ta._lib.BAD_ARG
    assert lib.tgp_fit(*args(c=-1.0)) == ta._lib.BAD_ARG
    assert lib.tgp_fit(*args(nls=3)) == ta._lib.BAD_ARG
    assert lib.tgp_fit(*args(noise=-1.0)) == ta._lib.BAD_ARG
    assert b"tgp_fit" in lib.tgp_last_error(h)
    assert lib.tgp_fit(h, None, 6, 2, p(y), 0, 1.0, p(ls), 1, 0.0, 0.0, 1, None, None, None) == ta._lib.BAD_ARG
    assert lib.tgp_fit(*args()) == ta._lib.OK
    assert lib.tgp_sweep(h, 3, -1.0, 0.0, 0.01, None, None, None, None, None, None) == ta._lib.BAD_ARG     # no candidates yet
    assert lib.tgp_set_candidates(h, p(X), 0) == ta._lib.BAD_ARG
    assert lib.tgp_set_candidates(h, p(X), 6) == ta._lib.OK
    assert lib.tgp_sweep(h, 9, -1.0, 0.0, 0.01, None, None, None, None, None, None) == ta._lib.BAD_ARG     # unknown acquisition
    assert lib.tgp_sweep(h, 3, 0.5, 0.0, 0.01, None, None, None, None, None, None) == ta._lib.BAD_ARG      # sf must be +-1
    out = np.empty(2)
    assert lib.tgp_get_candidate(h, 6, p(out)) == ta._lib.BAD_ARG
    assert lib.tgp_get_candidate(h, 5, p(out)) == ta._lib.OK and np.array_equal(out, X[5])
    idx = ctypes.c_int64(-1)
    assert lib.tgp_sweep(h, 3, -1.0, float(y.min()), 0.01, None, None, None, ctypes.byref(best), ctypes.byref(idx), None) == ta._lib.OK
    assert 0 <= idx.value < 6 and np.isfinite(best.value)
    # borrowed device pointers are checked against the runtime's records before any kernel sees them
    import torch
    vp = ctypes.c_void_p
    assert lib.tgp_set_candidates_dev(h, vp(X.ctypes.data), 6) == ta._lib.BAD_ARG          # host memory
    assert b"tgp_set_candidates_dev" in lib.tgp_last_error(h)
    dev = torch.from_numpy(X).to("cuda:0")
    # torch's caching allocator hands out pieces of 2 MiB segments: the runtime only knows the segment
    assert lib.tgp_set_candidates_dev(h, vp(dev.data_ptr()), 10 ** 7) == ta._lib.BAD_ARG   # 160 MB do not fit
    assert lib.tgp_set_candidates_dev(h, vp(dev.data_ptr() + 4), 2) == ta._lib.BAD_ARG     # misaligned
    assert lib.tgp_set_candidates_dev(h, vp(dev.data_ptr() + 16), 5) == ta._lib.OK         # rows 1..5 of the tensor
    assert lib.tgp_get_candidate(h, 0, p(out)) == ta._lib.OK and np.array_equal(out, X[1])
    assert lib.tgp_set_candidates_dev(h, vp(dev.data_ptr()), 6) == ta._lib.OK
    assert lib.tgp_sweep(h, 3, -1.0, float(y.min()), 0.01, None, None, None, ctypes.byref(best), ctypes.byref(idx), None) == ta._lib.OK
    assert lib.tgp_set_candidates(h, p(X), 6) == ta._lib.OK                                 # back to an owned copy
    del dev
    Xd = X.copy(); Xd[3] = Xd[1]                                            # singular without noise / jitter
    assert lib.tgp_fit(h, p(Xd), 6, 2, p(y), 0, 1.0, p(ls), 1, 0.0, 0.0, 1, None, None, None) == ta._lib.NOT_PD
    assert b"positive definite" in lib.tgp_last_error(h)
    assert lib.tgp_sweep(h, 3, -1.0, 0.0, 0.01, None, None, None, None, None, None) == ta._lib.NOT_FITTED   # a failed fit leaves no model
    assert lib.tgp_destroy(h) == ta._lib.OK
    assert lib.tgp_destroy(None) == ta._lib.OK


def test_large_n_vs_oracle(ta):
    """N = 9000 (Np = 9216: 18 outer Cholesky blocks, an odd tail in the inverse merges) against
    the oracle, f64: factor-level quantities and the sweep"""
    N, D, M = 9000, 6, 3000
    X, y, Xc = _synth(901, N, D, M)
    ls, noise = float(np.sqrt(D / 6.0)), 1e-3
    om = o.fit(X, y, "matern52", 1.0, ls, noise, 1e-10, True)
    gp = ta.NativeGP(0, "f64")
    lml, ym, ys = gp.fit(X, y, "matern52", 1.0, ls, noise, 1e-10, True)
    assert lml == pytest.approx(om.lml, rel=1e-9)
    np.testing.assert_allclose(gp.debug_read(ta._lib.BUF_ALPHA), om.alpha.ravel(), rtol=1e-6, atol=1e-7 * np.abs(om.alpha).max())
    gp.set_candidates(Xc)
    r = gp.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)
    mu, sg = o.predict(om, Xc, True)
    acq = o.acquisition("ei", mu, sg, "min", 0.01, float(y.min()))
    np.testing.assert_allclose(r["mu"], mu, rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(r["sigma"] ** 2, sg ** 2, rtol=1e-5, atol=1e-9 * (1.0 + noise) * ys ** 2)
    np.testing.assert_allclose(r["acq"], acq, rtol=1e-5, atol=1e-9)
    assert r["best_idx"] == int(np.argmax(acq))


def test_very_large_n_properties(ta):
    """N = 47000 (Np^2 > 2^31: every index product must be 64-bit; 4 x 17.7 GB of factor buffers),
    f32 sweep.  No oracle at this size; identities that hold for the exact GP instead:
    K alpha = yn on sampled rows, the posterior mean at training points, and the variance of the
    MFMA sweep against the independent GEMV path of tgp_acq_grad."""
    N, D, M = 47000, 5, 2048
    rng = np.random.RandomState(47)
    X = rng.uniform(0, 1, (N, D))
    y = np.sin(4 * X[:, 0]) + X[:, 1] * X[:, 2] + 0.05 * rng.normal(size=N)
    ls, noise, jit = 0.3, 1e-2, 1e-10
    gp = ta.NativeGP(0, "f32")
    lml, ym, ys = gp.fit(X, y, "rbf", 1.0, ls, noise, jit, True)
    assert np.isfinite(lml)
    alpha = gp.debug_read(ta._lib.BUF_ALPHA)
    yn = (y - ym) / ys
    rows = rng.choice(N, 200, replace=False)
    Krows = o.cross_kernel(X[rows], X, "rbf", 1.0, ls)           # (200, N), f64 on the host
    Krows[np.arange(200), rows] = 1.0 + noise + jit
    np.testing.assert_allclose(Krows @ alpha, yn[rows], rtol=0, atol=2e-7 * np.abs(alpha).max())
    # posterior mean at training points: mu_i = s_y (yn_i - (s2 + a) alpha_i) + ybar   (f32 sweep)
    Xq = np.vstack([X[rows], rng.uniform(0, 1, (M - 200, D))])
    gp.set_candidates(Xq)
    r = gp.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True)
    expect = ys * (yn[rows] - (noise + jit) * alpha[rows]) + ym
    np.testing.assert_allclose(r["mu"][:200], expect, rtol=0, atol=2e-3 * ys)
    assert np.all(r["sigma"] >= 0) and np.all(r["sigma"] <= np.sqrt(1.0 + noise) * ys * (1 + 1e-6))
    # variance: MFMA contraction (f32) against the f64 GEMV path on the same factor
    val, _ = gp.acq_grad(Xq[190:222], ta._lib.ACQ_SIGMA, 1.0, 0.0, 0.0)
    np.testing.assert_allclose(r["sigma"][190:222], val, rtol=2e-3, atol=2e-4 * ys)
    assert 0 <= r["best_idx"] < M


def test_two_handles_from_two_threads(ta):
    """one handle per thread, used concurrently (each on its own stream; the library keeps no
    shared mutable state beyond lock-free per-device flags): results equal the single-threaded ones"""
    import threading
    probs = []
    for seed, (N, D, M, kind, dtype) in enumerate([(700, 5, 30000, "matern52", "f32"), (450, 9, 20000, "rbf", "f64")]):
        X, y, Xc = _synth(300 + seed, N, D, M)
        probs.append((X, y, Xc, kind, dtype))

    def work(p, reps, out):
        X, y, Xc, kind, dtype = p
        gp = ta.NativeGP(0, dtype)
        res = []
        for _ in range(reps):
            lml, _, _ = gp.fit(X, y, kind, 1.0, 0.7, 1e-3, 1e-10, True)
            gp.set_candidates(Xc)
            r = gp.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_acq=True)
            res.append((lml, r["best_idx"], r["best_val"], r["acq"].copy()))
        out.append(res)

    ref = []
    for p in probs:
        work(p, 1, ref)
    outs = [[], []]
    threads = [threading.Thread(target=work, args=(probs[i], 6, outs[i])) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(2):
        assert len(outs[i]) == 1 and len(outs[i][0]) == 6
        for (lml, bi, bv, acq) in outs[i][0]:
            assert lml == ref[i][0][0] and bi == ref[i][0][1] and bv == ref[i][0][2]
            assert np.array_equal(acq, ref[i][0][3])


def test_no_device_memory_growth(ta):
    """workspaces are grow-only and owned by the handle: cycling fits / sweeps / gradients of
    mixed sizes must not leak device memory, and destroying the handle gives everything back"""
    import gc
    import torch
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    gp = ta.NativeGP(0, "f32")
    shapes = [(300, 5, 3000), (40, 2, 100), (700, 9, 20000), (129, 3, 1)]

    def cycle():
        for (N, D, M) in shapes:
            X, y, Xc = _synth(N, N, D, M)
            gp.fit(X, y, "matern52", 1.0, 0.7, 1e-3, 1e-10, True)
            gp.fit_grad(X, y, "matern52", 1.0, np.full(D, 0.7), 1e-3, 1e-10, True)
            gp.set_candidates(Xc)
            gp.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)
            gp.acq_grad(Xc[:3], ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01)
            gp.gen_candidates(1, 0, M, np.zeros(D), np.ones(D))
            gp.sweep(ta._lib.ACQ_UCB, -1.0, 0.0, 2.0)
            gp.import_state(gp.export_state())

    cycle()                                   # warm: every buffer reaches its largest size
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(0)
    for _ in range(5):
        cycle()
    torch.cuda.synchronize()
    free2, _ = torch.cuda.mem_get_info(0)
    assert free2 >= free1 - (4 << 20), "device memory shrank by %d bytes over 5 cycles" % (free1 - free2)
    del gp
    gc.collect()
    torch.cuda.synchronize()
    free3, _ = torch.cuda.mem_get_info(0)
    assert free3 >= free0 - (64 << 20), "handle destruction left %d bytes behind" % (free0 - free3)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_state_blob_round_trip(ta, dtype):
    """tgp_export_state / tgp_import_state (the persistence row of SURVEY 8f-4; what
    turbo/recorder.py:141-147 does with dill): a second handle rebuilt from the blob gives
    bitwise the same factor and sweep; the blob is O(N*D); damaged blobs are refused."""
    import ctypes
    X, y, Xc = _synth(77, 300, 5, 2000)
    ls = np.linspace(0.4, 1.1, 5)
    a = ta.NativeGP(0, dtype)
    lml, _, _ = a.fit(X, y, "matern52", 1.3, ls, 1e-3, 1e-10, True)
    blob = a.export_state()
    assert len(blob) == (8 + 5 + 300 * 5 + 300) * 8 and blob[:8] == b"TGPSTAT1"
    b = ta.NativeGP(0, dtype)
    assert b.import_state(blob) == lml
    assert np.array_equal(a.debug_read(ta._lib.BUF_LINV), b.debug_read(ta._lib.BUF_LINV))
    a.set_candidates(Xc); b.set_candidates(Xc)
    ra = a.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)
    rb = b.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)
    for k in ("mu", "sigma", "acq"):
        assert np.array_equal(ra[k], rb[k])
    assert ra["best_idx"] == rb["best_idx"]
    assert b.export_state() == blob
    # against the oracle built from the same (X, y, theta)
    om = o.fit(X, y, "matern52", 1.3, ls, 1e-3, 1e-10, True)
    np.testing.assert_allclose(lml, om.lml, rtol=1e-9)
    # refusals at the C boundary
    lib = ta._lib.load()
    buf = ctypes.create_string_buffer(blob, len(blob))
    vp = ctypes.cast(buf, ctypes.c_void_p)
    assert lib.tgp_import_state(b._h, vp, len(blob) - 8, None) == ta._lib.BAD_ARG     # truncated
    bad = ctypes.create_string_buffer(b"X" + blob[1:], len(blob))
    assert lib.tgp_import_state(b._h, ctypes.cast(bad, ctypes.c_void_p), len(blob), None) == ta._lib.BAD_ARG
    need = ctypes.c_int64()
    assert lib.tgp_export_state(a._h, vp, 16, ctypes.byref(need)) == ta._lib.BAD_ARG and need.value == len(blob)
    fresh = ta.NativeGP(0, dtype)
    assert lib.tgp_export_state(fresh._h, None, 0, ctypes.byref(need)) == ta._lib.NOT_FITTED


@pytest.mark.parametrize("N,D,M", [(2, 1, 1), (65, 3, 129), (257, 33, 1000), (130, 300, 70), (513, 2, 257)])
@pytest.mark.parametrize("kind", ["rbf", "matern12", "matern32", "matern52"])
def test_odd_shapes_vs_oracle(ta, N, D, M, kind):
    """tile edges everywhere: N, D, M around the 64 / 128 / 256 / 4 / 32 boundaries of the kernels"""
    X, y, Xc = _synth(900 + N + D, N, D, M)
    ls = np.sqrt(D / 6.0) * (0.6 + 0.8 * np.arange(D) / max(D - 1, 1)) if D > 1 else 0.4
    c, noise = 1.4, 2e-3
    om = o.fit(X, y, kind, c, ls, noise, 1e-10, True)
    omu, osig = o.predict(om, Xc)
    for dtype, tol in (("f64", 1e-7), ("f32", 5e-3)):
        gp = ta.NativeGP(0, dtype)
        lml, ym, ys = gp.fit(X, y, kind, c, ls, noise, 1e-10, True)
        assert lml == pytest.approx(om.lml, rel=1e-9, abs=1e-9)
        gp.set_candidates(Xc)
        r = gp.sweep(ta._lib.ACQ_UCB, 1.0, 0.0, 1.5, want_mu=True, want_sigma=True, want_acq=True)
        np.testing.assert_allclose(r["mu"], omu, rtol=tol, atol=tol * om.y_std)
        np.testing.assert_allclose(r["sigma"] ** 2, osig ** 2, rtol=tol, atol=tol * (c + noise) * om.y_std ** 2)
        assert r["best_idx"] == int(np.argmax(r["acq"]))
    if D <= 33:
        glml, ggrad = ta.NativeGP(0, "f64").fit_grad(X, y, kind, c, ls, noise, 1e-10, True)
        olml, ograd = o.lml_and_grad(X, y, kind, c, ls, noise, 1e-10, True)
        np.testing.assert_allclose(ggrad, ograd, rtol=1e-6, atol=1e-7 * max(1.0, np.abs(ograd).max()))


def test_seeded_fuzz_vs_oracle(ta):
    """150 seeded random problems (N 1..400, D 1..40, M 1..900, every kernel, iso / ARD, with and
    without normalisation, every acquisition and extremum) on ONE set of handles (f64, f32 and the
    two opt-in split-operand dtypes) that is reused throughout -- so workspaces shrink and grow
    between calls -- against the oracle"""
    rng = np.random.RandomState(20240601)
    gp64, gp32 = ta.NativeGP(0, "f64"), ta.NativeGP(0, "f32")
    gph2, gpx3 = ta.NativeGP(0, "f32h2"), ta.NativeGP(0, "f32x3")   # the opt-in split-operand sweeps: the f32 bound
    kinds = ["rbf", "matern12", "matern32", "matern52"]
    acqs = [("ucb", ta._lib.ACQ_UCB, 2.0), ("pi", ta._lib.ACQ_PI, 0.01), ("ei", ta._lib.ACQ_EI, 0.01)]
    for case in range(150):
        N = int(rng.choice([1, 2, 3, 5, 17, 63, 64, 65, 127, 128, 129, 200, 255, 256, 257, 300, 400]))
        D = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 31, 32, 33, 40]))
        M = int(rng.choice([1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 513, 900]))
        kind = kinds[rng.randint(4)]
        ard = bool(rng.randint(2)) and D > 1
        norm = bool(rng.randint(4))                    # mostly normalised, as the reference's default
        c = float(np.exp(rng.uniform(-1, 1)))
        base = np.sqrt(D / 6.0) * float(np.exp(rng.uniform(-0.5, 0.5)))
        ls = base * np.exp(rng.uniform(-0.4, 0.4, D)) if ard else base
        noise = float(10 ** rng.uniform(-5, -1))
        X = rng.uniform(-1, 2, (N, D))
        y = np.sin(X.sum(1)) * 3 + 10 + 0.1 * rng.normal(size=N)
        Xc = rng.uniform(-1, 2, (M, D))
        if M > 2 and N > 2:
            Xc[0] = X[0]                                # an observed point among the candidates
        name, acq, param = acqs[rng.randint(3)]
        ext = "max" if rng.randint(2) else "min"
        sf = 1.0 if ext == "max" else -1.0
        inc = float(y.max() if ext == "max" else y.min())
        tag = "case %d: N=%d D=%d M=%d %s ard=%s norm=%s %s/%s" % (case, N, D, M, kind, ard, norm, name, ext)
        om = o.fit(X, y, kind, c, ls, noise, 1e-10, norm)
        omu, osig = o.predict(om, Xc)
        oacq = o.acquisition(name, omu, osig, ext, param, inc)
        for gp, tol in ((gp64, 1e-7), (gp32, 5e-3), (gph2, 5e-3), (gpx3, 5e-3)):
            lml, ym, ys = gp.fit(X, y, kind, c, ls, noise, 1e-10, norm)
            assert lml == pytest.approx(om.lml, rel=1e-9, abs=1e-8), tag
            assert ym == pytest.approx(om.y_mean, rel=1e-14, abs=1e-14) and ys == pytest.approx(om.y_std, rel=1e-13), tag
            gp.set_candidates(Xc)
            r = gp.sweep(acq, sf, inc, param, want_mu=True, want_sigma=True, want_acq=True)
            np.testing.assert_allclose(r["mu"], omu, rtol=tol, atol=tol * max(om.y_std, 1e-3), err_msg=tag)
            np.testing.assert_allclose(r["sigma"] ** 2, osig ** 2, rtol=tol, atol=tol * (c + noise) * om.y_std ** 2, err_msg=tag)
            assert r["best_idx"] == int(np.argmax(r["acq"])), tag
            if gp is gp64:
                np.testing.assert_allclose(r["acq"], oacq, rtol=1e-5, atol=1e-9 * max(1.0, abs(inc)), err_msg=tag)
                assert r["acq"][r["best_idx"]] == pytest.approx(float(oacq.max()), rel=1e-6, abs=1e-9), tag


def test_seeded_fuzz_mid_sizes_vs_oracle(ta):
    """16 seeded random problems at the sizes the round-3 changes live at (N 400 .. 3100: fused and two-launch
    panel chains, both outer-block sizes, ragged last blocks; M up to 70 000: several slab groups per launch),
    every kernel, iso / ARD, f64 and f32 sweeps on handles reused throughout, against the oracle; the host
    backend answers the same model to 1e-9 of the oracle on a slice"""
    rng = np.random.RandomState(20261004)
    gp64, gp32 = ta.NativeGP(0, "f64"), ta.NativeGP(0, "f32")
    host = ta.NativeGP(ta._lib.DEVICE_HOST, "f64")
    kinds = ["rbf", "matern12", "matern32", "matern52"]
    acqs = [("ucb", ta._lib.ACQ_UCB, 2.0), ("pi", ta._lib.ACQ_PI, 0.01), ("ei", ta._lib.ACQ_EI, 0.01)]
    sizes = [401, 513, 640, 769, 1023, 1025, 1281, 1500, 1537, 1800, 2049, 2300, 2561, 2817, 3073, 3100]
    for case, N in enumerate(sizes):
        D = int(rng.choice([2, 5, 8, 13, 24, 33]))
        M = int(rng.choice([300, 4097, 20000, 70001]))
        kind = kinds[case % 4]
        ard = bool(rng.randint(2))
        c = float(np.exp(rng.uniform(-0.7, 0.7)))
        base = np.sqrt(D / 6.0) * float(np.exp(rng.uniform(-0.3, 0.3)))
        ls = base * np.exp(rng.uniform(-0.3, 0.3, D)) if ard else base
        noise = float(10 ** rng.uniform(-3.5, -1.5))
        X = rng.uniform(0, 1, (N, D))
        y = np.sin(3 * X @ rng.normal(size=D) / np.sqrt(D)) + 0.5 * ((X - 0.5) ** 2).sum(1) + 0.02 * rng.normal(size=N)
        Xc = rng.uniform(0, 1, (M, D))
        Xc[:2] = X[:2]
        name, acq, param = acqs[case % 3]
        ext = "max" if case % 2 else "min"
        sf = 1.0 if ext == "max" else -1.0
        inc = float(y.max() if ext == "max" else y.min())
        tag = "case %d: N=%d D=%d M=%d %s ard=%s %s/%s" % (case, N, D, M, kind, ard, name, ext)
        om = o.fit(X, y, kind, c, ls, noise, 1e-10, True)
        omu, osig = o.predict(om, Xc, True, chunk=8192)
        oacq = o.acquisition(name, omu, osig, ext, param, inc)
        for gp, tol in ((gp64, 1e-7), (gp32, 5e-3)):
            lml, ym, ys = gp.fit(X, y, kind, c, ls, noise, 1e-10, True)
            assert lml == pytest.approx(om.lml, rel=1e-9, abs=1e-8), tag
            if gp is gp64:
                np.testing.assert_allclose(gp.debug_read(ta._lib.BUF_L), om.L, rtol=1e-8, atol=1e-11, err_msg=tag)
            gp.set_candidates(Xc)
            r = gp.sweep(acq, sf, inc, param, want_mu=True, want_sigma=True, want_acq=True)
            np.testing.assert_allclose(r["mu"], omu, rtol=tol, atol=tol * max(om.y_std, 1e-3), err_msg=tag)
            np.testing.assert_allclose(r["sigma"] ** 2, osig ** 2, rtol=tol, atol=tol * (c + noise) * om.y_std ** 2, err_msg=tag)
            assert r["best_idx"] == int(np.argmax(r["acq"])), tag
            if gp is gp64:
                np.testing.assert_allclose(r["acq"], oacq, rtol=1e-5, atol=1e-9 * max(1.0, abs(inc)), err_msg=tag)
        if case % 4 == 0:
            host.fit(X, y, kind, c, ls, noise, 1e-10, True)
            rh = host.evaluate(Xc[:600], acq, sf, inc, param, True, True, True)
            np.testing.assert_allclose(rh["mu"], omu[:600], rtol=1e-9, atol=1e-9 * om.y_std, err_msg=tag)
            np.testing.assert_allclose(rh["sigma"] ** 2, osig[:600] ** 2, rtol=1e-7, atol=1e-9 * (c + noise) * om.y_std ** 2, err_msg=tag)


@pytest.mark.parametrize("env", [dict(TGP_SMALL="0"), dict(TGP_SMALL="0", TGP_PANEL="0")], ids=["blocked", "blocked-round1-panel"])
def test_golden_cases_on_the_blocked_path(env):
    """the golden cases are all small (N <= 64): by default they run on the small-problem kernels;
    here they also pin the blocked multi-launch path at the same sizes"""
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "_golden_child.py")],
                         env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "golden-child ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.parametrize("env", [
    dict(TGP_BGINV="0"), dict(TGP_BG_CUS="0"), dict(TGP_BG_CUS="64"),   # inverse level by level after the factorisation (default: block rows behind the panel chain on a 192-CU background stream); that stream unmasked / on 64 CUs
    dict(TGP_PANEL="3"), dict(TGP_PANEL="38"), dict(TGP_PANEL="8"), dict(TGP_PANEL="4"), dict(TGP_PANEL="0"),   # every diagonal-block factorisation variant (default: 5)
    dict(TGP_TRAIL64="0", TGP_MERGE64="0", TGP_INNER="gemm64"),          # 128-tile direct-to-LDS GEMMs everywhere in the fit
    dict(TGP_TRAIL64="100000", TGP_MERGE64="100000", TGP_OB="256"),      # 64-tile template everywhere, smaller outer block
    dict(TGP_TILE="128", TGP_CHUNK="1024"),                              # small sweep tiles, many launches
    dict(TGP_TILE="256x128", TGP_NBUF="2"),                              # big tiles forced, two LDS buffers
    dict(TGP_TRMM="reg"),                                                # register-staged sweep kernel
    dict(TGP_PANEL_FUSE="0"),                                            # the two-launch panel chain (default: one fused launch per panel up to Np = 4096)
    dict(TGP_SLAB_GB="40", TGP_CHUNK="2048"),                            # the whole batch in one launch pair, several slab groups inside it
    dict(TGP_SLAB_GB="0.001", TGP_CHUNK="1024", TGP_KS_JS="1"),          # one group per launch pair, unsplit cross-kernel grid
    dict(TGP_GEMM64="reg", TGP_LINV_ZERO="1", TGP_PANEL_FUSE_TILES="128", TGP_OB="512"),   # round 3's fit: register-staged 64-tile template, Linv zero-filled by every fit
    dict(TGP_MID="0", TGP_SWEEP_ZC="0"),                                 # 128 < N <= 256 down the general four-launch sweep; results by D2H copies + memset
    dict(TGP_MEAN="0", TGP_KS_JS="2"),                                   # round 4's posterior mean: partial sums in the cross-kernel instead of the contraction
    dict(TGP_LEVEL64_FUSED="0", TGP_BGINV="0"),                          # the inverse's 64 -> 128 level as three launches, level by level everywhere
], ids=["inverse-by-levels", "bg-unmasked", "bg-64cu", "panel-c4", "panel-c8", "panel-b8", "panel-b4", "panel-a", "fit-glds", "fit-64", "sweep-128", "sweep-256x128", "sweep-reg", "panel-unfused", "sweep-one-launch", "sweep-launch-per-group", "fit-round3", "no-mid-sweep", "mean-in-kstar", "level64-unfused"])
def test_alternate_kernel_paths(env):
    """every kernel selection the TGP_* switches offer stays correct: the
    defaults pick by size, so some variants would otherwise only run at sizes the suite never uses"""
    import subprocess
    import sys
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "_alt_paths_child.py")],
                         env=e, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "alt-paths ok" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
