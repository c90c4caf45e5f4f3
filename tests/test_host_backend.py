"""The reload path without a GPU (SURVEY.md 8b "Pickling", 8f-4): libturbogp.so's HOST backend,
tgp_create(TGP_DEVICE_HOST), against the reference-generated golden vectors, and a pickled model
reloaded in a process that sees no HIP device (HIP_VISIBLE_DEVICES="").  CPU only.

The host backend is product code of its own (csrc/host_backend.cpp: plain C++); the oracle is not
involved in what is measured here -- the expected values are the reference's own outputs
(tests/golden/*.npz)."""
import os
import pickle
import subprocess
import sys
import warnings

import numpy as np
import pytest

from conftest import ROOT, golden_path

ACQS = {"ei": (3, 0.01), "pi": (2, 0.01), "ucb2": (1, 2.0), "ucbinf": (4, 0.0)}


def _host():
    from turbo_amd import _lib
    return _lib, _lib.NativeGP(_lib.DEVICE_HOST, "f64")


def test_fit_state_vs_reference(golden_case):
    c = golden_case
    lib, gp = _host()
    lml, ym, ys = gp.fit(c["X"], c["y"], str(c["kind"]), float(c["constant"]), c["length_scale"],
                         float(c["noise"]), float(c["jitter"]), bool(c["normalize_y"]))
    assert ym == pytest.approx(float(c["y_mean"]), rel=1e-13, abs=1e-14)
    assert ys == pytest.approx(float(c["y_std"]), rel=1e-13)
    np.testing.assert_allclose(gp.debug_read(lib.BUF_L), c["L"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(gp.debug_read(lib.BUF_ALPHA), c["alpha"], rtol=1e-6, atol=1e-8)
    assert lml == pytest.approx(float(c["lml"]), rel=1e-9, abs=1e-8)
    # the state blob round-trips on the host handle and rebuilds the same model
    blob = gp.export_state()
    lib2, gp2 = _host()
    assert gp2.import_state(blob) == lml


@pytest.mark.parametrize("ext", ["min", "max"])
def test_predict_and_acquisitions_vs_reference(golden_case, ext):
    c = golden_case
    lib, gp = _host()
    gp.fit(c["X"], c["y"], str(c["kind"]), float(c["constant"]), c["length_scale"],
           float(c["noise"]), float(c["jitter"]), bool(c["normalize_y"]))
    scale = (float(c["constant"]) + float(c["noise"])) * float(c["y_std"]) ** 2
    for name, (acq, param) in ACQS.items():
        sf = 1.0 if ext == "max" else -1.0
        r = gp.evaluate(c["Xc"], acq, sf, float(c["incumbent_" + ext]), param, True, True, True)
        want = c["acq_%s_%s" % (name, ext)]
        np.testing.assert_allclose(r["mu"], c["mus"], rtol=1e-9, atol=1e-10)
        np.testing.assert_allclose(r["sigma"] ** 2, c["sigmas"] ** 2, rtol=1e-7, atol=1e-9 * scale)
        # the acquisition inherits sigma's cancellation floor near observed points (as tests/test_gpu_parity.py)
        s_floor = np.sqrt(1e-9 * scale)
        big = max(1.0, float(np.abs(want).max()))
        well = c["sigmas"] > 100 * s_floor
        np.testing.assert_allclose(r["acq"][well], want[well], rtol=1e-6, atol=1e-9 * big)
        np.testing.assert_allclose(r["acq"], want, rtol=1e-5, atol=(param + 2) * 2 * s_floor + 1e-9 * big)
        assert r["acq"][r["best_idx"]] == r["acq"].max() and r["best_idx"] == int(np.argmax(r["acq"]))
        assert r["best_val"] == r["acq"][r["best_idx"]]


def test_not_pd_and_gpu_only_entries():
    lib, gp = _host()
    with np.load(golden_path("not_pd"), allow_pickle=False) as z:
        with pytest.raises(np.linalg.LinAlgError):
            gp.fit(z["X"], z["y"], "rbf", 1.0, z["length_scale"], 0.0, 0.0, True)
    gp.fit(np.array([[0.0], [1.0]]), np.array([1.0, 3.0]), "rbf", 1.0, 0.5, 0.0, 1e-10, True)
    with pytest.raises(ValueError, match="host backend"):
        gp.fit_grad(np.array([[0.0], [1.0]]), np.array([1.0, 3.0]), "rbf", 1.0, 0.5, 0.0, 1e-10, True)
    with pytest.raises(ValueError, match="host backend"):
        gp.gen_candidates(1, 0, 10, [0.0], [1.0])


def test_larger_model_threads_agree():
    """N above the panel width and the thread thresholds: the threaded Cholesky and sweep equal the
    single-thread run bit for bit (shares are rows / candidate tiles: no cross-thread sums)"""
    code = r'''
import sys, hashlib, numpy as np
sys.path.insert(0, %r)
from turbo_amd import _lib
rng = np.random.RandomState(5)
N, D, M = 700, 6, 900
X = rng.uniform(0, 1, (N, D)); y = np.sin(3 * X.sum(1)) + 0.05 * rng.normal(size=N); Xc = rng.uniform(0, 1, (M, D))
gp = _lib.NativeGP(_lib.DEVICE_HOST, "f64")
lml, _, _ = gp.fit(X, y, "matern32", 1.2, 0.8, 1e-3, 1e-10, True)
r = gp.evaluate(Xc, _lib.ACQ_EI, -1.0, float(y.min()), 0.01, True, True, True)
h = hashlib.sha256(gp.debug_read(_lib.BUF_L).tobytes() + r["mu"].tobytes() + r["sigma"].tobytes() + r["acq"].tobytes()).hexdigest()
print(repr(lml), r["best_idx"], h)
''' % ROOT
    outs = []
    for t in ("1", "5"):
        e = dict(os.environ, TGP_HOST_THREADS=t)
        outs.append(subprocess.check_output([sys.executable, "-c", code], env=e, timeout=300).decode())
    assert outs[0] == outs[1]


_CHILD = r'''
import sys, pickle, warnings, numpy as np
sys.path.insert(0, %r)
import turbo_amd as ta
blob = open(sys.argv[1], "rb").read()
with warnings.catch_warnings(record=True) as ws:
    warnings.simplefilter("always")
    model, acq_args = pickle.loads(blob)
    Xc = np.load(sys.argv[2])
    mu, sg = model.predict(Xc, return_std_dev=True)
    out = dict(mu=mu, sigma=sg, lml=np.array(model.get_log_likelihood()), names=np.array(model.get_hyper_param_names()),
               hp=model.get_hyper_params())
    for name, (cls, param, ext, inc) in acq_args.items():
        fac = getattr(ta, cls)(param)
        f, _ = (fac.construct_function(0, model, ext, inc) if fac.get_type() == "improvement"
                else fac.construct_function(0, model, ext))
        out[name] = f(Xc)
assert any("host backend" in str(w.message) for w in ws), [str(w.message) for w in ws]
assert model._factory._context().host
np.savez(sys.argv[3], **out)
# a factory made HERE (not reloaded) must still refuse to run without a GPU
try:
    ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("rbf", 1.0, 1.0, 1e-3), optimizer=None), training_iterations=1
                      ).construct_model(0, np.zeros((3, 2)) + np.arange(3)[:, None], np.arange(3.0))
except ta._lib.NoDeviceError:
    print("fresh factory refused")
print("reload-child ok")
'''


@pytest.mark.parametrize("case", ["rbf_iso_8d", "matern52_ard_16d", "default_matern52_white_branin", "matern12_iso_4d_nowhite"])
def test_pickled_model_reloads_without_a_gpu(tmp_path, case):
    """what Recorder.load_compressed + the plot path do in another process (turbo/recorder.py:157-163,
    turbo/plotting/trials.py:192-195, :574-577): unpickle a trial's model where no HIP device is
    visible, predict a grid and evaluate the acquisitions -- equal to the reference's own outputs"""
    import turbo_amd as ta
    with np.load(golden_path(case), allow_pickle=False) as z:
        c = {k: z[k] for k in z.files}
    ls = c["length_scale"]
    kern = ta.GPKernel(str(c["kind"]), float(c["constant"]), ls if np.ndim(ls) and len(ls) > 1 else float(np.ravel(ls)[0]),
                       float(c["noise"]) if float(c["noise"]) > 0 else None)
    fac = ta.HipGPSurrogate(model_params=dict(kernel=kern, optimizer=None, alpha=float(c["jitter"]),
                                              normalize_y=bool(c["normalize_y"])), training_iterations=1)
    # (what construct_model returns, minus the fit that needs the GPU: the pickle holds host state only)
    model = ta.HipGPSurrogate.ModelInstance(fac, np.array(c["X"]), np.array(c["y"]), kern.copy(), float(c["jitter"]),
                                            bool(c["normalize_y"]))
    acq_args = {"ei_min": ("EI", 0.01, "min", float(c["incumbent_min"])), "pi_max": ("PI", 0.01, "max", float(c["incumbent_max"])),
                "ucb2_min": ("UCB", 2.0, "min", None), "ucbinf_max": ("UCB", float("inf"), "max", None)}
    pk, xc, out = tmp_path / "model.pkl", tmp_path / "xc.npy", tmp_path / "out.npz"
    pk.write_bytes(pickle.dumps((model, acq_args)))
    np.save(xc, c["Xc"])
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    res = subprocess.run([sys.executable, "-c", _CHILD % ROOT, str(pk), str(xc), str(out)], env=env,
                         capture_output=True, text=True, timeout=300)
    assert res.returncode == 0 and "reload-child ok" in res.stdout and "fresh factory refused" in res.stdout, \
        res.stdout[-2000:] + res.stderr[-4000:]
    with np.load(out, allow_pickle=False) as z:
        np.testing.assert_allclose(z["mu"], c["mus"], rtol=1e-9, atol=1e-10)
        scale = (float(c["constant"]) + float(c["noise"])) * float(c["y_std"]) ** 2
        np.testing.assert_allclose(z["sigma"] ** 2, c["sigmas"] ** 2, rtol=1e-7, atol=1e-9 * scale)
        assert float(z["lml"]) == pytest.approx(float(c["lml"]), rel=1e-9, abs=1e-8)
        np.testing.assert_allclose(z["hp"], c["hyper_params"], rtol=1e-12)
        assert [str(s) for s in z["names"]] == [str(s) for s in c["hyper_param_names"]]
        s_floor = np.sqrt(1e-9 * scale)
        well = c["sigmas"] > 100 * s_floor
        for name, (cls, param, ext, inc) in acq_args.items():
            want = c["acq_%s_%s" % (name.split("_")[0], ext)]
            big = max(1.0, float(np.abs(want).max()))
            np.testing.assert_allclose(z[name][well], want[well], rtol=1e-6, atol=1e-9 * big, err_msg=name)
            np.testing.assert_allclose(z[name], want, rtol=1e-5, atol=((param if np.isfinite(param) else 1.0) + 2) * 2 * s_floor + 1e-9 * big, err_msg=name)


def test_host_only_library_has_no_rocm_dependency_and_serves_the_reload_path(tmp_path):
    """libturbogp_host.so: the same host backend as a library of its own for machines WITHOUT ROCm (the
    full library links libamdhip64.so and cannot be loaded there).  It must not depend on any ROCm /
    HIP library, must export only what a TGP_DEVICE_HOST handle serves, and turbo_amd must fall back to
    it when libturbogp.so is not there to load -- answering the golden case like the full library."""
    import turbo_amd as ta
    from turbo_amd import _lib
    host = _lib.HOST_LIB_PATH
    assert os.path.exists(host), "make -C turbo_amd/csrc builds libturbogp_host.so beside libturbogp.so"
    needed = subprocess.check_output(["readelf", "-d", host]).decode()
    assert "amdhip" not in needed and "hsa" not in needed and "rocm" not in needed.lower(), needed
    syms = subprocess.check_output(["nm", "-D", "--defined-only", host]).decode()
    exported = {l.split()[-1] for l in syms.splitlines() if " T " in l and l.split()[-1].startswith("tgp_")}   # the C entries
    assert {"tgp_create", "tgp_fit", "tgp_evaluate", "tgp_predict", "tgp_sweep", "tgp_import_state"} <= exported
    assert not ({"tgp_fit_grad", "tgp_sweep_topk", "tgp_acq_refine", "tgp_multi_create"} & exported)
    assert exported <= set(_lib.SYMBOLS)                    # nothing the header does not declare
    with np.load(golden_path("rbf_iso_8d"), allow_pickle=False) as z:
        c = {k: z[k] for k in z.files}
    kern = ta.GPKernel("rbf", float(c["constant"]), float(np.ravel(c["length_scale"])[0]), float(c["noise"]))
    fac = ta.HipGPSurrogate(model_params=dict(kernel=kern, optimizer=None, alpha=float(c["jitter"]), normalize_y=True),
                            training_iterations=1)
    model = ta.HipGPSurrogate.ModelInstance(fac, np.array(c["X"]), np.array(c["y"]), kern.copy(), float(c["jitter"]), True)
    pk, xc, out = tmp_path / "m.pkl", tmp_path / "xc.npy", tmp_path / "out.npz"
    pk.write_bytes(pickle.dumps(model))
    np.save(xc, c["Xc"])
    child = r"""
import sys, pickle, warnings, numpy as np
sys.path.insert(0, %r)
import turbo_amd as ta
from turbo_amd import _lib
warnings.simplefilter("ignore")
m = pickle.loads(open(sys.argv[1], "rb").read())
mu, sg = m.predict(np.load(sys.argv[2]), return_std_dev=True)
assert _lib.HOST_ONLY and m._factory._context().host
assert "broken_libturbogp.so" in _lib.LOAD_ERROR          # the dlopen message is kept ...
try:
    ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel(), optimizer=None), training_iterations=1)
    raise SystemExit("a new factory out of the host-only library?")
except _lib.TurboGPLibraryError as e:
    assert "broken_libturbogp.so" in str(e)               # ... and shown where a new factory is refused
try:
    _lib.NativeGP(0)
    raise SystemExit("a GPU context out of the host-only library?")
except _lib.NoDeviceError:
    pass
try:
    m._factory._context().fit_grad(m.X, m.y, "rbf", 1.0, 1.0, 1e-3, 1e-10, True)
    raise SystemExit("a GPU-only entry out of the host-only library?")
except _lib.TurboGPLibraryError:
    pass
np.savez(sys.argv[3], mu=mu, sg=sg)
print("host-only ok", _lib.load().tgp_version().decode())
""" % ROOT
    # the full library "cannot be loaded" (as on a machine without the ROCm runtime it links): TGP_LIBRARY names
    # a file that is there but is no shared object
    broken = tmp_path / "broken_libturbogp.so"
    broken.write_bytes(b"not an ELF file")
    env = dict(os.environ, TGP_LIBRARY=str(broken))
    res = subprocess.run([sys.executable, "-c", child, str(pk), str(xc), str(out)], env=env, capture_output=True,
                         text=True, timeout=300)
    assert res.returncode == 0 and "host-only ok" in res.stdout and "host-only" in res.stdout.split("ok", 1)[1], \
        res.stdout[-2000:] + res.stderr[-4000:]
    with np.load(out, allow_pickle=False) as z:
        np.testing.assert_allclose(z["mu"], c["mus"], rtol=1e-9, atol=1e-10)
        scale = (float(c["constant"]) + float(c["noise"])) * float(c["y_std"]) ** 2
        np.testing.assert_allclose(z["sg"] ** 2, c["sigmas"] ** 2, rtol=1e-7, atol=1e-9 * scale)
    # an explicit TGP_LIBRARY that does not exist is a mistake, not a reason to settle for the host-only build
    env = dict(os.environ, TGP_LIBRARY=str(tmp_path / "no_such_libturbogp.so"))
    res = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from turbo_amd import _lib; _lib.load()" % ROOT],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and "does not exist" in res.stderr


def test_host_backend_under_sanitizers(tmp_path):
    """AddressSanitizer + UBSan over the host backend (fit with the threaded panel Cholesky, sweep on
    16-candidate tiles, the state blob round trip) at sizes either side of the panel / tile widths"""
    csrc = os.path.join(ROOT, "turbo_amd", "csrc")
    exe = str(tmp_path / "host_san")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-fno-omit-frame-pointer", "-pthread", "-I" + csrc, os.path.join(ROOT, "tests", "host_sanitizer_driver.cpp"),
           os.path.join(csrc, "host_backend.cpp"), "-o", exe]
    built = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if built.returncode != 0 and ("asan" in built.stderr.lower() or "sanitize" in built.stderr.lower()):
        pytest.skip("this g++ has no sanitizer runtime: " + built.stderr[-200:])
    assert built.returncode == 0, built.stderr[-3000:]
    run = subprocess.run([exe], env=dict(os.environ, TGP_HOST_THREADS="5"), capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "ERROR" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-3000:]
    lines = run.stdout.strip().splitlines()
    assert len(lines) == 7 and all(" rc=0 " in l and "lml2==lml 1" in l for l in lines), run.stdout
