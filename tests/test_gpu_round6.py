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


def test_stream_status_and_the_warning_when_the_hardware_queues_ran_out():
    """tgp_stream_status: what the probes of the device's shared streams said.  In a process with ONE hardware queue
    (GPU_MAX_HW_QUEUES=1 before the runtime initialises) every stream runs behind every other: the probes say so,
    NativeGP warns once, and everything still computes.  In the default process the streams overlap and nothing warns."""
    code = r'''
import sys, warnings
sys.path.insert(0, %r)
import numpy as np
%s
import turbo_amd as ta
with warnings.catch_warnings(record=True) as ws:
    warnings.simplefilter("always")
    gp = ta.NativeGP(0, "f64")
    gp2 = ta.NativeGP(0, "f64")          # the warning is given once per device
st = gp.stream_status()
n = sum("SERIALISED" in str(w.message) for w in ws)
rng = np.random.RandomState(0)
X = rng.uniform(0, 1, (700, 3)); y = np.sin(X.sum(1))
lml, _, _ = gp.fit(X, y, "rbf", 1.0, 0.7, 1e-2, 1e-10, True)
print("status", st["background_overlaps"], st["third_overlaps"], st["gpu_max_hw_queues"], "warnings", n, "lml", repr(lml))
'''
    outs = {}
    # (a CU-masked stream is a hardware queue of its own whatever GPU_MAX_HW_QUEUES says: the one-queue process also
    # asks for UNMASKED background / third streams, which the runtime deals onto its one queue)
    for name, env, pre in (("default", {}, ""), ("one-queue", dict(GPU_MAX_HW_QUEUES="1", TGP_BG_CUS="0", TGP_PRE_CUS="0"), ""),
                           ("torch-first", {}, "import torch; torch.zeros(4, device='cuda').sum().item()")):
        e = dict(os.environ)
        e.pop("GPU_MAX_HW_QUEUES", None)
        e.update(env)
        out = subprocess.run([sys.executable, "-c", code % (ROOT, pre)], env=e, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0 and "status" in out.stdout, (name, out.stdout[-500:], out.stderr[-3000:])
        outs[name] = out.stdout.strip().splitlines()[-1].split()
    d, o, t = outs["default"], outs["one-queue"], outs["torch-first"]
    assert d[1] == "1" and d[2] == "1" and d[3] == "8" and d[5] == "0", d          # both overlap, turbo_amd set 8 queues, no warning
    assert o[1] == "0" and o[2] == "0" and o[3] == "1" and o[5] == "1", o          # serialised, said so once
    assert d[7] == o[7] == t[7], (d, o, t)                                         # the same fit either way
    # torch first: the runtime was initialised with its default 4 queues (turbo_amd's setdefault came too late but is
    # what the environment shows); whether the three shared streams still found queues of their own is the runtime's
    # deal -- the status and the warning must agree
    assert (t[1] == "0" or t[2] == "0") == (t[5] == "1"), t


def test_winner_wait_orders_another_stream_behind_the_record():
    import torch
    import turbo_amd as ta
    rng = np.random.RandomState(4)
    X = rng.uniform(0, 1, (300, 4))
    y = np.sin(3 * X.sum(1))
    gp = ta.NativeGP(0, "f64")
    gp.fit(X, y, "matern52", 1.0, 0.8, 1e-3, 1e-10, True)
    with pytest.raises(ValueError):
        gp.winner_wait(None)                       # no record attached
    Xc = rng.uniform(0, 1, (5000, 4))
    gp.set_candidates(Xc)
    rec = torch.zeros(6, dtype=torch.float64, device="cuda:0")
    gp.set_winner_out(rec.data_ptr(), 1000, keepalive=rec)
    gp.winner_wait(None)                           # attached, no sweep yet: nothing to wait for
    side = torch.cuda.Stream()
    r = gp.sweep(ta._lib.ACQ_EI, -1.0, float(y.min()), 0.01)
    gp.winner_wait(side.cuda_stream)
    with torch.cuda.stream(side):
        got = rec.clone()                          # issued on the side stream, behind the event
    side.synchronize()
    got = got.cpu().numpy()
    assert got[0] == r["best_val"] and int(got[1]) == 1000 + r["best_idx"]
    np.testing.assert_array_equal(got[2:], Xc[r["best_idx"]])


@pytest.mark.parametrize("N,D,kind,dtype,M", [(40, 3, "matern52", "f64", 3000), (100, 4, "rbf", "f32", 3000),
                                             (200, 5, "matern32", "f32", 4000), (700, 6, "matern52", "f64", 5000),
                                             (1300, 8, "rbf", "f32", 6000), (700, 6, "rbf", "f32h2", 5000)])
def test_a_handle_sweeps_with_a_factor_it_received(N, D, kind, dtype, M):
    """tgp_export_factor_dev / tgp_import_factor_dev: handle B receives what a sweep needs of handle A's fit -- whole,
    and 128 rows at a time -- and returns A's sweep bit for bit: every mean, deviation and acquisition value, the winner,
    the acquisition gradient; it is not fitted before the last block is in, it refuses to export a training set it does
    not have, and a fit of its own afterwards is a fit again"""
    import turbo_amd as ta
    L = ta._lib
    rng = np.random.RandomState(N)
    X = rng.uniform(0, 1, (N, D))
    y = np.sin(3 * X.sum(1)) + 0.3 * ((X - 0.5) ** 2).sum(1) + 0.02 * rng.normal(size=N)
    Xc = rng.uniform(0, 1, (M, D))
    ls = float(np.sqrt(D / 6.0))
    a = ta.NativeGP(0, dtype)
    lml, ym, ys = a.fit(X, y, kind, 1.1, ls, 1e-2, 1e-10, True)
    a.set_candidates(Xc)
    ra = a.sweep(L.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)
    ga = a.acq_grad(Xc[:5], L.ACQ_EI, -1.0, float(y.min()), 0.01)
    f = a.export_factor()
    assert (f.N, f.D, f.Np) == (N, D, -(-N // 256) * 256) and f.lml == lml and f.y_mean == ym
    for blocks in (None, 128):
        b = ta.NativeGP(0, dtype)
        if blocks is None:
            assert b.import_factor(f)
        else:
            for r0 in range(0, f.Np, blocks):
                done = b.import_factor(f, r0, blocks)
                assert done == (r0 + blocks == f.Np)
                if not done:
                    with pytest.raises(RuntimeError):      # not fitted until the last row block has arrived
                        b.acq_grad(Xc[:2], L.ACQ_EI, -1.0, float(y.min()), 0.01)
            with pytest.raises(ValueError):
                b.import_factor(f, 128, 128)           # out of order: a block other than the next one
        b.set_candidates(Xc)
        rb = b.sweep(L.ACQ_EI, -1.0, float(y.min()), 0.01, want_mu=True, want_sigma=True, want_acq=True)
        for k in ("mu", "sigma", "acq"):
            np.testing.assert_array_equal(rb[k], ra[k])
        assert rb["best_idx"] == ra["best_idx"] and rb["best_val"] == ra["best_val"]
        gb = b.acq_grad(Xc[:5], L.ACQ_EI, -1.0, float(y.min()), 0.01)
        np.testing.assert_array_equal(gb[0], ga[0])
        np.testing.assert_array_equal(gb[1], ga[1])
        with pytest.raises(ValueError):
            b.export_state()
        # ... and a fit of its own afterwards is an ordinary fit (the buffers a receiver did without are allocated now)
        lml_b, _, _ = b.fit(X, y, kind, 1.1, ls, 1e-2, 1e-10, True)
        assert lml_b == lml
        np.testing.assert_array_equal(b.debug_read(L.BUF_LINV), a.debug_read(L.BUF_LINV))
        b.close()
    a.close()


def test_the_matrix_core_products_of_tgp_acq_grad_against_the_earlier_kernels_and_the_oracle():
    """tgp_acq_grad above N = 128: z = Linv k and w = Linv^T z on v_mfma_f64_16x16x4 (default) against the wave-per-row /
    split-column kernels (TGP_QUERY_MFMA=0, a child process: csrc/tuning.hpp is read once) -- the same values and
    gradients to rounding at three size classes (4 / 8 / 16 waves per block), 1 ... 40 points (three groups of sixteen),
    and a point's result bit for bit the same alone and inside a batch"""
    code = r'''
import sys, json
import numpy as np
sys.path.insert(0, %r)
import turbo_amd as ta
out = []
for N, D in ((300, 3), (1000, 6), (1500, 5)):
    rng = np.random.RandomState(N)
    X = rng.uniform(0, 1, (N, D)); y = np.sin(3 * X.sum(1)) + 0.01 * rng.normal(size=N)
    gp = ta.NativeGP(0, "f64")
    gp.fit(X, y, "matern52", 1.0, float(np.sqrt(D / 6.0)), 1e-3, 1e-10, True)
    P = rng.uniform(0, 1, (40, D)); P[3] = X[5]
    for acq, par in ((ta._lib.ACQ_EI, 0.01), (ta._lib.ACQ_UCB, 2.0)):
        v, g = gp.acq_grad(P, acq, -1.0, float(y.min()), par)
        v1, g1 = gp.acq_grad(P[17:18], acq, -1.0, float(y.min()), par)
        assert v1[0] == v[17] and np.array_equal(g1[0], g[17])
        v9, g9 = gp.acq_grad(P[:9], acq, -1.0, float(y.min()), par)
        assert np.array_equal(v9, v[:9]) and np.array_equal(g9, g[:9])
        out.append([v.tolist(), g.tolist()])
print("RESULT" + json.dumps(out))
''' % ROOT
    import json
    res = {}
    for name, env in (("mfma", {}), ("old", dict(TGP_QUERY_MFMA="0"))):
        e = dict(os.environ)
        e.update(env)
        r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "RESULT" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
        res[name] = json.loads(r.stdout.split("RESULT")[1])
    for a, b in zip(res["mfma"], res["old"]):
        va, ga, vb, gb = np.array(a[0]), np.array(a[1]), np.array(b[0]), np.array(b[1])
        assert np.allclose(va, vb, rtol=1e-9, atol=1e-13), np.max(np.abs(va - vb))
        assert np.allclose(ga, gb, rtol=1e-7, atol=1e-10), np.max(np.abs(ga - gb))
