"""Round 6 (GPU): the short calls of the everyday regime -- small fit, fit + LML gradient in one launch, acquisition value
+ gradient in one launch, all with a polled completion (csrc/doorbell.hpp) -- against round 5's calls on the same build,
bit for bit where the switch promises the same bytes; the one-launch hyper-parameter fit twice (the regression recorded
in profiles/r06_device_optimiser_bisect.txt); what tgp_last_timings says about the last sweep's arithmetic."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIGEST = os.path.join(ROOT, "tools", "short_calls_digest.py")


def _digest(env=None, skip=""):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, DIGEST] + (["--skip", skip] if skip else []), env=e, capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 0 and "digest done" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
    return [ln for ln in out.stdout.splitlines() if ln and not ln.startswith("digest done")]


@pytest.fixture(scope="module")
def default_lines():
    return _digest()


def test_the_short_calls_return_the_same_bytes_twice(default_lines):
    """every short call of a fresh process, default kernels: the same lines again (the one-launch optimiser included:
    its first round-6 form returned a different optimum on every run)"""
    again = _digest()
    assert again == default_lines, "\n".join(a + "\n" + b for a, b in zip(default_lines, again) if a != b)
    assert sum(ln.startswith("device") for ln in default_lines) == 10


def test_round5_body_of_the_small_fit_returns_the_same_bytes(default_lines):
    """TGP_SMALL_LIVE=0: the pivot chain walks the padding blocks, the targets are fetched in front of alpha (round 5's
    body) -- the fit, both gradients, the library's L-BFGS-B, the one-launch optimiser and the acquisition gradients on
    top of that fit: the same bytes"""
    legacy = _digest(dict(TGP_SMALL_LIVE="0"))
    assert legacy == default_lines, "\n".join(a + "\n" + b for a, b in zip(default_lines, legacy) if a != b)


def test_round5_calls_return_the_same_bytes(default_lines):
    """TGP_POLL_US=0 TGP_SMALL_FUSED=0: events + hipStreamSynchronize, fit and gradient as two launches, tgp_acq_grad
    through pageable copies (round 5's calls).  Same bytes for everything but tgp_acq_grad at N <= 128, which then runs
    the general kernels (another order of the sums: the same values to rounding, compared below)"""
    old = _digest(dict(TGP_POLL_US="0", TGP_SMALL_FUSED="0"))
    keep = lambda lines: [ln for ln in lines if not (ln.startswith("acq_grad") and "general" not in ln)]
    assert keep(old) == keep(default_lines), "\n".join(a + "\n" + b for a, b in zip(keep(default_lines), keep(old)) if a != b)
    # the general query kernels polled (value + gradient formed by the reduction's last workgroup, written to mapped
    # host memory) against copied (q_finalize_kernel + D2H): the same bytes
    assert [ln for ln in old if "general" in ln] == [ln for ln in default_lines if "general" in ln]


def test_one_launch_query_against_the_general_kernels_and_the_oracle():
    """tgp_acq_grad for N <= 128 (small_query_kernel: a workgroup per point) against the general kernels on the same
    handle's fit (TGP_SMALL_QUERY is read per process: the general kernels are reached through m > the staging limit
    here? no -- through a child) and against finite differences of the oracle's acquisition"""
    import turbo_amd as ta
    from oracle import gp_oracle as o
    L = ta._lib
    rng = np.random.RandomState(5)
    for N, D, kind in ((7, 2, "matern52"), (40, 3, "rbf"), (64, 5, "matern32"), (100, 4, "matern52"), (128, 6, "matern12")):
        X = rng.uniform(0, 1, (N, D))
        y = np.sin(3 * X.sum(1)) + 0.3 * ((X - 0.4) ** 2).sum(1) + 0.02 * rng.normal(size=N)
        ls = float(np.sqrt(D / 6.0))
        gp = ta.NativeGP(0, "f64")
        gp.fit(X, y, kind, 1.3, ls, 2e-2, 1e-10, True)
        om = o.fit(X, y, kind, 1.3, ls, 2e-2, 1e-10, True)
        P = rng.uniform(0.05, 0.95, (9, D))
        for acq, name, par in ((L.ACQ_EI, "ei", 0.01), (L.ACQ_PI, "pi", 0.01), (L.ACQ_UCB, "ucb", 2.0)):
            v, g = gp.acq_grad(P, acq, -1.0, float(y.min()), par)
            want, _, _ = o.sweep(om, P, name, "min", par, float(y.min()))
            np.testing.assert_allclose(v, want, rtol=1e-7, atol=1e-12)
            h = 1e-6
            for d in range(D):
                Pp, Pm = P.copy(), P.copy()
                Pp[:, d] += h
                Pm[:, d] -= h
                fd = (o.sweep(om, Pp, name, "min", par, float(y.min()))[0] - o.sweep(om, Pm, name, "min", par, float(y.min()))[0]) / (2 * h)
                np.testing.assert_allclose(g[:, d], fd, rtol=2e-4, atol=1e-7 * max(1.0, np.abs(fd).max()))


def test_polled_calls_survive_a_spin_limit_that_always_expires():
    """TGP_POLL_US=1: the host gives up on the doorbell almost at once and synchronises the stream instead -- the
    fall-back every polled call has; same results as the default"""
    a = _digest(skip="lbfgsb,device")
    b = _digest(dict(TGP_POLL_US="1"), skip="lbfgsb,device")
    assert a == b


def test_last_timings_say_which_arithmetic_the_last_sweep_ran_in():
    import turbo_amd as ta
    rng = np.random.RandomState(2)
    gp = ta.NativeGP(0, "f32")
    assert gp.last_timings()["sweep_f64"] == -1
    for N, want in ((40, 1), (200, 1), (700, 0)):
        X = rng.uniform(0, 1, (N, 4))
        y = np.sin(X.sum(1))
        gp.fit(X, y, "rbf", 1.0, 0.8, 1e-2, 1e-10, True)
        gp.set_candidates(rng.uniform(0, 1, (3000, 4)))
        gp.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01)
        assert gp.last_timings()["sweep_f64"] == want, (N, gp.last_timings())


def test_sweep_dtype_asks_the_library_when_the_one_launch_sweep_is_switched_off():
    """an f32 factory at N = 200 with TGP_MID=0: the general sweep runs in f32, sweep_dtype says so, and CandidateSweep
    re-forms the winner's value in float64 (round-5 advisor: the size rule alone said 'f64')"""
    code = r'''
import numpy as np, sys
sys.path.insert(0, %r)
import turbo_amd as ta
rng = np.random.RandomState(3)
X = rng.uniform(0, 1, (200, 3)); y = np.sin(3 * X.sum(1))
sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("rbf", 1.0, 0.7, 1e-2), optimizer=None, normalize_y=True),
                        training_iterations=1, dtype="f32")
model, _ = sur.construct_model(0, X, y)
f, _ = ta.EI(0.01).construct_function(0, model, "min", float(y.min()))
b = ta.Bounds([("x%%d" %% d, 0.0, 1.0) for d in range(3)])
np.random.seed(0)
x, info = ta.CandidateSweep(num_random=4000)(b, f)
print("dtype", f.sweep_dtype, "resweep", "max_acq_sweep" in info)
''' % ROOT
    for env, want in ((dict(TGP_MID="0"), "dtype f32 resweep True"), (dict(), "dtype f64 resweep False")):
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and want in out.stdout, (env, out.stdout[-500:], out.stderr[-2000:])
