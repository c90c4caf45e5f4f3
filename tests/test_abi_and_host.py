"""CPU-only: the C-ABI library loads and exports every symbol include/turbogp.h declares (no
compute calls), argument validation that needs no GPU, and the host-side plugin logic."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _lib():
    import __graft_entry__ as g
    import turbo_amd
    if not os.path.exists(turbo_amd.LIB_PATH):
        g.build()
    return turbo_amd._lib


def test_header_symbols_are_exported():
    lib = _lib()
    h = open(os.path.join(ROOT, "include", "turbogp.h")).read()
    declared = set(re.findall(r"\b(tgp_[a-z_0-9]+)\s*\(", h))
    assert declared == set(lib.SYMBOLS), declared ^ set(lib.SYMBOLS)
    cdll = lib.load()
    for name in declared:
        assert hasattr(cdll, name), name
    assert cdll.tgp_version().decode().startswith("turbogp")
    # the shared object exports exactly the declared C symbols (plus nothing tgp_-prefixed)
    nm = subprocess.run(["nm", "-D", "--defined-only", lib.LIB_PATH], stdout=subprocess.PIPE, text=True).stdout
    exported = set(re.findall(r"\bT (tgp_[a-z_0-9]+)\b", nm))
    assert exported == declared, exported ^ declared


def _build_c_consumer(tmp_path):
    exe = str(tmp_path / "c_abi_consumer")
    csrc = os.path.join(ROOT, "turbo_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O1",
                           os.path.join(ROOT, "tests", "c_abi_consumer.c"), "-o", exe,
                           "-L" + csrc, "-lturbogp", "-lm", "-Wl,-rpath," + csrc])
    return exe


def test_header_is_plain_c_and_links_from_c(tmp_path):
    """include/turbogp.h compiles as C99 (-pedantic) and a C program links and calls the library"""
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only",
                           "-x", "c", os.path.join(ROOT, "include", "turbogp.h")])
    out = subprocess.check_output([_build_c_consumer(tmp_path)], timeout=60).decode()
    assert "c-abi ok" in out


@pytest.mark.gpu
def test_c_consumer_on_the_gpu(tmp_path):
    """the same C program doing a fit + sweep + state round trip, checked against closed forms"""
    out = subprocess.check_output([_build_c_consumer(tmp_path), "--gpu"], timeout=120).decode()
    assert "c-abi ok (gpu)" in out


def test_library_has_gfx950_code_object():
    lib = _lib()
    blob = open(lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob


def test_built_kernels_wait_for_their_lds_reads_before_a_barrier():
    """What the compiler made of the sources, read from libturbogp.so's gfx950 code objects: in every kernel that
    stages operands global -> LDS directly, no s_barrier is reachable -- along ANY path of the control-flow graph,
    loop back-edges included -- with LDS reads pending (tools/check_lds_dma_barriers.py: that wait once sat BELOW the
    barrier in the f64 GEMM's k-loop and fits came out wrong under load).  The shipped library holds no known-racy
    loop any more; the checker proves it is not blind on tools/microbench/lds_race_demo.hip, which it compiles itself:
    the pre-fix k-loop must offend, the shipped one must pass."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_lds_dma_barriers.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and ", 0 offending" in r.stdout, r.stdout[-3000:] + r.stderr[-2000:]
    assert "the pre-fix k-loop offends, the shipped k-loop passes -> the checker is not blind" in r.stdout, r.stdout[-3000:]


def test_the_shipped_library_holds_no_debug_instantiation_of_the_f64_gemm():
    """round 4 shipped gemm64_glds_kernel<.., DBG = 1..5> -- among them the k-loop known to return wrong tiles -- behind
    TGP_GEMM64=round4-war; they now exist only under -DTGP_DEBUG_KERNELS (make debug -> libturbogp_dbg.so, tools/microbench)"""
    out = subprocess.run(["nm", "-C", os.path.join(ROOT, "turbo_amd", "csrc", "libturbogp.so")], capture_output=True, text=True, check=True).stdout
    inst = set(re.findall(r"gemm64_glds_kernel<(\d+), (\d+), (\d+), (\d+)>", out))
    assert inst and all(nbuf == "3" and dbg == "0" for _, _, nbuf, dbg in inst), sorted(inst)


def test_every_environment_switch_is_in_the_one_table():
    """csrc/tuning.hpp is the ONLY place the library reads its environment (DESIGN.md section 5 is printed from it): no
    getenv elsewhere in csrc/ outside the debug build's block, and tgp_tuning() lists every TGP_* name the sources,
    tests and tools use"""
    csrc = os.path.join(ROOT, "turbo_amd", "csrc")
    stray = []
    for f in sorted(os.listdir(csrc)):
        if not f.endswith((".hip", ".hpp", ".cpp")) or f == "tuning.hpp":
            continue
        src = open(os.path.join(csrc, f)).read()
        src = re.sub(r"#ifdef TGP_DEBUG_KERNELS.*?#endif", "", src, flags=re.S)
        stray += ["%s: %s" % (f, m) for m in re.findall(r".*getenv\(.*", src)]
    assert not stray, stray
    import turbo_amd
    table = turbo_amd._lib.tuning()
    assert len(table) >= 30 and all(k.startswith("TGP_") and doc for k, (v, doc) in table.items())
    used = set()
    for d in ("tests", "tools", os.path.join("tools", "gpu")):
        for f in os.listdir(os.path.join(ROOT, d)):
            if f.endswith((".py", ".sh")):
                used |= set(re.findall(r"\bTGP_[A-Z0-9_]+\b", open(os.path.join(ROOT, d, f)).read()))
    # names that are not switches of the library: enum constants of the C-ABI, the Python binding's own variables,
    # the build macro, test-harness variables
    not_switches = {"TGP_DEVICE_HOST", "TGP_LIBRARY", "TGP_HIP_RUNTIME", "TGP_NO_DEVICE", "TGP_DEBUG_KERNELS", "TGP_STRESS_QUICK",
                    "TGP_F64", "TGP_F32", "TGP_OK", "TGP_NOT_PD", "TGP_BAD_ARG", "TGP_RBF", "TGP_ACQ_EI", "TGP_ACQ_NONE", "TGP_ACQ_UCB",
                    "TGP_ACQ_PI", "TGP_ACQ_SIGMA", "TGP_MATERN52", "TGP_MATERN32", "TGP_MATERN12", "TGP_NOT_FITTED", "TGP_HIP_ERROR",
                    "TGP_NO_MEMORY", "TGP_F32X3", "TGP_F32H2", "TGP_"}
    missing = sorted(n for n in used - not_switches if n not in table)
    assert not missing, missing


def test_create_fails_loudly_without_gpu():
    lib = _lib()
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(lib.TurboGPLibraryError):
        lib.NativeGP(0, "f64")


def test_missing_library_is_a_loud_error(tmp_path, monkeypatch):
    import turbo_amd
    lib = turbo_amd._lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    host = lib.HOST_LIB_PATH
    monkeypatch.setattr(lib, "HOST_LIB_PATH", str(tmp_path / "nope_host.so"))
    with pytest.raises(lib.TurboGPLibraryError):
        lib.load()
    with pytest.raises(lib.TurboGPLibraryError):
        turbo_amd.HipGPSurrogate()
    # only the host-only build there (a machine without ROCm): loading works -- reloaded models need it --
    # but a NEW factory, i.e. the optimisation path, still refuses loudly, and so does a GPU context
    monkeypatch.setattr(lib, "HOST_LIB_PATH", host)
    monkeypatch.setattr(lib, "HOST_ONLY", False)
    assert lib.load() is not None and lib.HOST_ONLY
    with pytest.raises(lib.TurboGPLibraryError, match="host-only"):
        turbo_amd.HipGPSurrogate()
    with pytest.raises(lib.NoDeviceError):
        lib.NativeGP(0)


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "turbo_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, os.path.join(dirpath, f)


# ---- kernel specification ----------------------------------------------------------------------

def test_gpkernel_names_and_params():
    import turbo_amd as ta
    k = ta.GPKernel("rbf", 2.0, 0.5, 1e-3)
    assert k.hyper_param_names() == ["k1__k1__constant_value", "k1__k2__length_scale", "k2__noise_level"]
    np.testing.assert_array_equal(k.hyper_params(), [2.0, 0.5, 1e-3])
    k = ta.GPKernel("matern52", 1.0, [0.5, 0.7, 0.9], None)
    assert k.hyper_param_names() == ["k1__constant_value", "k2__length_scale_0", "k2__length_scale_1",
                                     "k2__length_scale_2"]
    np.testing.assert_array_equal(k.hyper_params(), [1.0, 0.5, 0.7, 0.9])
    with pytest.raises(AssertionError):
        ta.GPKernel("cubic")
    with pytest.raises(AssertionError):
        ta.GPKernel("rbf", -1.0)


def test_gpkernel_from_sklearn_objects():
    sk = pytest.importorskip("sklearn.gaussian_process.kernels")
    import turbo_amd as ta
    k = ta.GPKernel.from_any(1.0 * sk.Matern(nu=2.5) + sk.WhiteKernel())     # the reference default
    assert (k.kind, k.constant, k.length_scale, k.noise) == ("matern52", 1.0, 1.0, 1.0)
    skk = sk.ConstantKernel(2.5) * sk.Matern(length_scale=np.array([0.3, 0.4]), nu=1.5) + sk.WhiteKernel(1e-2)
    k = ta.GPKernel.from_any(skk)
    assert k.kind == "matern32" and k.constant == 2.5 and k.noise == 1e-2
    np.testing.assert_array_equal(k.length_scale, [0.3, 0.4])
    # same naming / ordering as the reference wrapper reads from sklearn (surrogates.py:340-362)
    names = []
    params = skk.get_params()
    for h in skk.hyperparameters:
        p = params[h.name]
        names.extend(["{}_{}".format(h.name, i) for i in range(len(p))] if isinstance(p, np.ndarray) else [h.name])
    assert k.hyper_param_names() == names
    np.testing.assert_allclose(k.hyper_params(), np.exp(skk.theta), rtol=1e-14)
    # fixed hyper-parameters are dropped from the values, like the reference does
    kf = ta.GPKernel.from_any(sk.ConstantKernel(1.0, "fixed") * sk.RBF(0.7) + sk.WhiteKernel(1e-4, "fixed"))
    np.testing.assert_array_equal(kf.hyper_params(), [0.7])
    assert ta.GPKernel.from_any(sk.RBF(0.3)).hyper_param_names() == ["length_scale"]
    for bad in (sk.RBF(1.0) + sk.RBF(2.0), sk.Matern(nu=0.7), sk.DotProduct(), sk.RBF(1.0) * sk.WhiteKernel()):
        with pytest.raises(ValueError):
            ta.GPKernel.from_any(bad)


# ---- auxiliary optimiser host logic -----------------------------------------------------------

class _FakeAcq:
    """stands in for a native function instance: records the batch, returns a chosen arg-max"""

    def __init__(self, f):
        self.f = f
        self.seen = None

    def maximise(self, X):
        self.seen = X
        v = self.f(X)
        i = int(np.argmax(v))
        return i, float(v[i])


def test_candidate_sweep_contract():
    import turbo_amd as ta
    b = ta.Bounds([("a", -5.0, 10.0), ("b", 0.0, 15.0)])
    acq = _FakeAcq(lambda X: -((X[:, 0] - 1.0) ** 2 + (X[:, 1] - 2.0) ** 2))
    np.random.seed(3)
    x, info = ta.CandidateSweep(num_random=4000)(b, acq)
    assert x.shape == (1, 2) and acq.seen.shape == (4000, 2)
    assert set(info) == {"max_acq"} and isinstance(info["max_acq"], float)
    assert abs(x[0, 0] - 1.0) < 0.5 and abs(x[0, 1] - 2.0) < 0.5
    # same draws as the reference's random_selector: one uniform column per parameter
    np.random.seed(3)
    c0 = np.random.uniform(-5.0, 10.0, size=(4000, 1))
    c1 = np.random.uniform(0.0, 15.0, size=(4000, 1))
    np.testing.assert_array_equal(acq.seen, np.hstack([c0, c1]))
    # a foreign callable goes through the reference's argsort path; NaN never wins
    vals = np.array([0.1, np.nan, 0.7, 0.7, -1.0])
    x, info = ta.CandidateSweep(num_random=5, gen_random=lambda n, lb: np.arange(10.0).reshape(5, 2))(
        b, lambda X: vals)
    assert info["max_acq"] == 0.7 and x.tolist() == [[4.0, 5.0]]
    # clipping to the bounds (auxiliary_optimisers.py:120-124)
    x, _ = ta.CandidateSweep(num_random=1, gen_random=lambda n, lb: np.array([[11.0, -1.0]]))(
        b, _FakeAcq(lambda X: np.zeros(len(X))))
    assert x.tolist() == [[10.0, 0.0]]
    # the gradient stage (auxiliary_optimisers.py:69-112) with a foreign callable: SciPy's own
    # finite differences, like the reference
    np.random.seed(4)
    quad = lambda X: -((X[:, 0] - 1.0) ** 2 + (X[:, 1] - 2.0) ** 2)
    x, info = ta.CandidateSweep(num_random=50, grad_restarts=4, start_from_best=2)(b, quad)
    np.testing.assert_allclose(x, [[1.0, 2.0]], atol=1e-5)
    assert info["max_acq"] == pytest.approx(0.0, abs=1e-9)
    # ... and with an instance that supplies value_and_grad (what the native instances do)

    class _Grad(_FakeAcq):
        calls = 0

        def __call__(self, X):
            return self.f(X)

        def value_and_grad(self, X):
            _Grad.calls += 1
            return self.f(X), np.stack([-2 * (X[:, 0] - 1.0), -2 * (X[:, 1] - 2.0)], axis=1)
    x, info = ta.CandidateSweep(num_random=50, grad_restarts=3, start_from_best=1)(b, _Grad(quad))
    np.testing.assert_allclose(x, [[1.0, 2.0]], atol=1e-6)
    assert _Grad.calls > 0
    with pytest.raises(AssertionError):
        ta.CandidateSweep(num_random=10, grad_restarts=2, start_from_best=3)
    assert issubclass(ta.RandomAndQuasiNewton, ta.CandidateSweep)      # (the reference's name, with the reference's defaults)
    # lock-step restarts: same answer as one after the other, far fewer gradient calls
    bumpy = lambda X: np.sin(3 * X[:, 0]) * np.cos(2 * X[:, 1]) - 0.01 * ((X[:, 0] - 2) ** 2 + (X[:, 1] - 7) ** 2)

    class _Bumpy(_FakeAcq):
        calls = 0
        points = 0

        def __call__(self, X):
            return self.f(X)

        def value_and_grad(self, X):
            _Bumpy.calls += 1
            _Bumpy.points += len(X)
            g0 = 3 * np.cos(3 * X[:, 0]) * np.cos(2 * X[:, 1]) - 0.02 * (X[:, 0] - 2)
            g1 = -2 * np.sin(3 * X[:, 0]) * np.sin(2 * X[:, 1]) - 0.02 * (X[:, 1] - 7)
            return self.f(X), np.stack([g0, g1], axis=1)
    res = {}
    for mode in (True, False):
        np.random.seed(11)
        _Bumpy.calls = _Bumpy.points = 0
        sweep = ta.CandidateSweep(num_random=64, grad_restarts=8, start_from_best=3, lockstep=mode)
        x, info = sweep(b, _Bumpy(bumpy))
        res[mode] = (x.copy(), info["max_acq"], _Bumpy.calls, _Bumpy.points, sweep.last_batches)
    np.testing.assert_array_equal(res[True][0], res[False][0])
    assert res[True][1] == res[False][1]
    assert res[True][3] == res[False][3]                     # every restart evaluated the same points
    assert res[True][2] < res[False][2] / 3                  # ... in far fewer calls
    assert res[True][4][0] == 8 and sum(res[True][4]) == res[True][3]
    # an evaluator that fails takes every restart down with the same exception, no deadlock

    class _Boom(_Bumpy):
        def value_and_grad(self, X):
            raise FloatingPointError("boom")
    with pytest.raises(FloatingPointError):
        ta.CandidateSweep(num_random=8, grad_restarts=4, start_from_best=0)(b, _Boom(bumpy))


def test_candidate_sweep_topk_edge_cases():
    """start_from_best beyond the library's top-k width (64) uses every one of the best rows, as the
    reference's argsort()[:start_from_best] does (turbo/modules/auxiliary_optimisers.py:63-66, :77-79);
    a batch that ranks nothing (all NaN) still yields grad_restarts starts and index 0"""
    import turbo_amd as ta
    b = ta.Bounds([("a", 0.0, 4.0), ("b", 0.0, 4.0)])
    quad = lambda X: -((X[:, 0] - 1.0) ** 2 + (X[:, 1] - 2.0) ** 2)

    class _TopK:
        """a native-like instance: top-k from the 'device' (k <= 64), refine on the 'device'"""
        def __init__(self, f, nan=False):
            self.f, self.nan, self.topk_calls, self.starts = f, nan, 0, None

        def __call__(self, X):
            return np.full(len(X), np.nan) if self.nan else self.f(X)

        def maximise_topk(self, X, k):
            self.topk_calls += 1
            assert k <= 64
            if self.nan:
                return np.empty(0, dtype=np.int64), np.empty(0)
            v = self.f(X)
            order = np.argsort(-v, kind="stable")[:k]
            return order, v[order]

        def refine(self, starting_points, bounds, max_iter=200):
            self.starts = np.array(starting_points)
            return self.starts, self.f(self.starts), 1

    np.random.seed(3)
    cand = np.random.uniform(0, 4, (500, 2))
    gen_calls = []

    def gen(n, lb):
        gen_calls.append(n)
        return cand[:n] if n == 500 else np.random.uniform(0, 4, (n, 2))
    acq = _TopK(quad)
    x, info = ta.CandidateSweep(num_random=500, grad_restarts=120, start_from_best=100, gen_random=gen, on_device=True)(b, acq)
    assert acq.topk_calls == 0                       # k > 64: the vector is argsorted on the host
    assert acq.starts.shape == (120, 2)
    best100 = cand[np.argsort(-quad(cand), kind="stable")[:100]]
    np.testing.assert_array_equal(acq.starts[:100], best100)      # ALL 100 best rows start a restart
    assert gen_calls == [500, 20]
    acq = _TopK(quad)
    ta.CandidateSweep(num_random=500, grad_restarts=12, start_from_best=10, gen_random=gen, on_device=True)(b, acq)
    assert acq.topk_calls == 1 and acq.starts.shape == (12, 2)
    np.testing.assert_array_equal(acq.starts[:10], best100[:10])
    # nothing ranked: no IndexError, the missing starts are drawn at random, the winner is index 0
    gen_calls.clear()
    acq = _TopK(quad, nan=True)
    x, info = ta.CandidateSweep(num_random=500, grad_restarts=6, start_from_best=4, gen_random=gen, on_device=True)(b, acq)
    assert acq.starts.shape == (6, 2) and gen_calls == [500, 6]
    assert np.isfinite(x).all()


def test_acquisition_factories_host_side():
    import turbo_amd as ta

    class M:   # enough of a native model for the host-side logic
        def _sweep(self, X, acq, sf, inc, param, **kw):
            self.args = (acq, sf, inc, param, kw)
            return {"acq": np.zeros(len(X)), "best_idx": 0, "best_val": 0.0}
    m = M()
    f, info = ta.UCB(beta=lambda t: 0.5 * t).construct_function(6, m, "min")
    assert info == {"beta": 3.0} and f.get_name() == "-LCB" and ta.UCB(1).get_type() == "optimism"
    f(np.zeros((2, 1)))
    assert m.args[:4] == (ta._lib.ACQ_UCB, -1, 0.0, 3.0)
    f, _ = ta.UCB(float("inf")).construct_function(0, m, "max")
    f.maximise(np.zeros((2, 1)))
    assert m.args[0] == ta._lib.ACQ_SIGMA and f.get_name() == "UCB"
    f, info = ta.EI(xi=0.01).construct_function(0, m, "max", 3.5)
    f(np.zeros((2, 1)))
    assert info == {"xi": 0.01} and m.args[:4] == (ta._lib.ACQ_EI, 1, 3.5, 0.01)
    assert ta.EI(0).get_type() == ta.PI(0).get_type() == "improvement"
    f, _ = ta.PI(xi=lambda t: 0.1).construct_function(0, m, "min", -2.0)
    f(np.zeros((2, 1)))
    assert m.args[:4] == (ta._lib.ACQ_PI, -1, -2.0, 0.1) and f.get_name() == "PI"
    with pytest.raises(AssertionError):
        ta.EI(0.01).construct_function(0, m, "sideways", 0.0)


def test_surrogate_argument_checks_need_no_gpu():
    import turbo_amd as ta
    _lib()
    with pytest.raises(AssertionError):
        ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel(), n_restarts_optimizer=3), training_iterations=2)
    s = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel(), optimizer=None),
                          training_iterations=lambda t: [10, 5, 2][t % 3])
    assert [s._get_training_iterations(t) for t in range(4)] == [10, 5, 2, 10]
    with pytest.raises(AssertionError):
        ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel()))._get_training_iterations(0)
    with pytest.raises(AssertionError):
        ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel()), training_iterations=-1)._get_training_iterations(0)


def test_philox_reference_known_answers():
    """Random123 known-answer vectors for Philox-4x32-10: pins the NumPy reference the GPU
    generator is compared against"""
    from philox_ref import philox4x32_10
    got = [int(v) for v in philox4x32_10(0, 0, 0, 0, 0, 0)]
    assert got == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    got = [int(v) for v in philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff)]
    assert got == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    got = [int(v) for v in philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822, 0x299f31d0)]
    assert got == [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_random_and_quasi_newton_keeps_the_references_defaults():
    """turbo/modules/auxiliary_optimisers.py:17: RandomAndQuasiNewton(num_random=1000, grad_restarts=10, start_from_best=2);
    CandidateSweep() is the pure sweep"""
    import inspect
    import turbo_amd as ta
    r = ta.RandomAndQuasiNewton()
    assert (r.num_random, r.grad_restarts, r.start_from_best) == (1000, 10, 2) and isinstance(r, ta.CandidateSweep)
    c = ta.CandidateSweep()
    assert (c.num_random, c.grad_restarts, c.start_from_best) == (1000, 0, 0)
    r2 = ta.RandomAndQuasiNewton(5000, 4, 1, on_device=True)
    assert (r2.num_random, r2.grad_restarts, r2.start_from_best, r2.on_device) == (5000, 4, 1, True)
    ref = "/root/reference/turbo/modules/auxiliary_optimisers.py"
    if os.path.exists(ref):          # (the build container only: held to the reference's own signature)
        import re
        m = re.search(r"class RandomAndQuasiNewton:\s+def __init__\(self, ([^)]*)\)", open(ref).read())
        assert m and m.group(1).replace(" ", "") == "num_random=1000,grad_restarts=10,start_from_best=2"
        assert list(inspect.signature(ta.RandomAndQuasiNewton.__init__).parameters)[1:4] == ["num_random", "grad_restarts", "start_from_best"]
