"""GPU parity at BASELINE.json's FULL sizes, on bench.py's own synthetic inputs:

    C1  8D RBF,            N=512,  M=65 536,    UCB beta=2, f64   oracle on ALL candidates
    C2  16D Matern-5/2 ARD, N=2048, M=131 072,   EI,         f64   oracle on ALL candidates
    C3  32D RBF,           N=4096, M=262 144,   EI,         f32   oracle on ALL candidates + exact regret
    C4  64D Matern-3/2,    N=8192, M=1 048 576, PI,         f32   whole batch on one GPU, a 131 072
                                                                  shard (the 8-GPU share), oracle on
                                                                  16 384 candidates + regret

Everything goes through the C-ABI (plugin classes -> ctypes -> libturbogp.so).  The oracle is the
checker only.  fp64 bar: rtol 1e-5 (north_star).  fp32 bar (the 1e-5 target is stated for fp64
only): |d mu| <= 5e-4 s_y, |d var| <= 5e-5 (c + s2) s_y^2 -- about 7x the deviation measured on
these inputs -- and an arg-max regret below 1e-3 relative, where regret is
judged by the f64 ORACLE's acquisition values.
"""
import json
import os

import numpy as np
import pytest

from oracle import gp_oracle as o

pytestmark = pytest.mark.gpu

RTOL = 1e-5
VAR_ATOL = 1e-9
F32_MU_TOL = 5e-4        # x y_std
F32_VAR_TOL = 5e-5       # x (c + noise) * y_std^2
REGRET_TOL = 1e-3

ACQ_CLS = {"ucb": "UCB", "pi": "PI", "ei": "EI"}
_measured = {}


@pytest.fixture(scope="module")
def ta():
    import turbo_amd
    return turbo_amd


@pytest.fixture(scope="module", autouse=True)
def _dump_measured():
    yield
    # measured deviations of this run, for DESIGN.md (scratch; not read by any test)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "config_parity_measured.json"), "w") as fh:
            json.dump(_measured, fh, indent=1, sort_keys=True)
    except OSError:
        pass


def _inputs(name, m=None):
    import bench
    cfg = dict(bench.CONFIGS[name])
    X, y, Xc, ls = bench.synth(cfg, 0, cfg["M"] if m is None else m)
    return cfg, X, y, Xc, ls


def _model(ta, cfg, X, y, ls, dtype=None):
    sur = ta.HipGPSurrogate(
        model_params=dict(kernel=ta.GPKernel(cfg["kind"], 1.0, ls, cfg["noise"]), optimizer=None,
                          normalize_y=True, alpha=1e-10),
        training_iterations=1, dtype=dtype or cfg["dtype"])
    model, _ = sur.construct_model(0, X, y)
    return sur, model


def _acq(ta, cfg, model, y):
    fac = getattr(ta, ACQ_CLS[cfg["acq"]])(cfg["param"])
    args = [0, model, "min"] + ([float(y.min())] if fac.get_type() == "improvement" else [])
    return fac.construct_function(*args)[0]


def _oracle(cfg, X, y, ls):
    return o.fit(X, y, cfg["kind"], 1.0, ls, cfg["noise"], 1e-10, True)


def _oracle_acq(cfg, om, Xs, y, chunk=16384):
    mu, sg = o.predict(om, Xs, True, chunk=chunk)
    return mu, sg, o.acquisition(cfg["acq"], mu, sg, "min", cfg["param"], float(y.min()))


def _common_properties(ta, cfg, sur, model, f, X, y, Xc, full, rng):
    """size-independent properties the domain offers (rows are independent given the fit)"""
    M = Xc.shape[0]
    noise = cfg["noise"]
    # sharding invariance: two uneven shards == the whole
    h = (M // 3) | 1
    np.testing.assert_array_equal(np.concatenate([f(Xc[:h]), f(Xc[h:])]), full)
    # arg-max == arg-max of the vector (lowest index on ties), idempotent
    bi, bv = f.maximise(Xc)
    assert bi == int(np.argmax(full)) and bv == full[bi]
    assert f.maximise(Xc) == (bi, bv)
    # permutation equivariance on a slice
    perm = rng.permutation(4096)
    np.testing.assert_array_equal(f(Xc[:4096][perm]), full[:4096][perm])
    # the posterior mean interpolates: mu(x_i) = s_y (yn_i - (s2 + a) alpha_i) + ybar
    alpha = sur._context().debug_read(ta._lib.BUF_ALPHA)
    yn = (y - model.y_mean) / model.y_std
    expect = model.y_std * (yn - (noise + 1e-10) * alpha) + model.y_mean
    tol = 1e-7 if sur.dtype == "f64" else F32_MU_TOL
    np.testing.assert_allclose(model.predict(X), expect, rtol=0, atol=tol * model.y_std)
    # UCB is linear in beta
    mu, sg = model.predict(Xc[:8192], return_std_dev=True)
    u, _ = ta.UCB(2.0).construct_function(0, model, "min")
    np.testing.assert_allclose(u(Xc[:8192]), -mu + 2.0 * sg, rtol=1e-14, atol=1e-14)
    return bi, bv


def _check_f64(cfg, om, got_mu, got_sg, got_acq, omu, osg, oacq, tag):
    np.testing.assert_allclose(got_mu, omu, rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(got_sg ** 2, osg ** 2, rtol=RTOL,
                               atol=VAR_ATOL * (1.0 + cfg["noise"]) * om.y_std ** 2)
    np.testing.assert_allclose(got_acq, oacq, rtol=RTOL, atol=1e-12 + 1e-9 * float(np.abs(oacq).max()))
    _measured[tag] = dict(
        mu_rel=float(np.max(np.abs(got_mu - omu)) / om.y_std),
        var_rel=float(np.max(np.abs(got_sg ** 2 - osg ** 2)) / ((1.0 + cfg["noise"]) * om.y_std ** 2)),
        acq_abs=float(np.max(np.abs(got_acq - oacq))))


def _check_f32(cfg, om, got_mu, got_sg, omu, osg, tag):
    emu = float(np.max(np.abs(got_mu - omu)) / om.y_std)
    evar = float(np.max(np.abs(got_sg ** 2 - osg ** 2)) / ((1.0 + cfg["noise"]) * om.y_std ** 2))
    _measured[tag] = dict(mu_rel=emu, var_rel=evar)
    assert emu < F32_MU_TOL, (tag, emu)
    assert evar < F32_VAR_TOL, (tag, evar)


def _regret(cfg, om, y, Xc, f32_choice, ranking, n_top, n_rand, rng, tag):
    """arg-max regret of the f32 sweep, judged by the f64 oracle: the oracle's acquisition at the
    candidate the f32 sweep chose against the oracle's best over (the n_top best candidates of
    `ranking` + n_rand random ones + the choice).  `ranking` is an f64 GPU acquisition vector; it
    only proposes where the oracle should look, the values compared are the oracle's."""
    top = np.argsort(-ranking, kind="stable")[:n_top]
    rand = rng.choice(Xc.shape[0], n_rand, replace=False)
    idx = np.unique(np.concatenate([top, rand, [f32_choice]]))
    _, _, oacq = _oracle_acq(cfg, om, Xc[idx], y, chunk=8192)
    best = float(oacq.max())
    chosen = float(oacq[int(np.searchsorted(idx, f32_choice))])
    regret = (best - chosen) / max(abs(best), 1e-300)
    _measured[tag] = dict(regret=regret, oracle_best=best, oracle_at_choice=chosen, checked=int(len(idx)))
    assert regret < REGRET_TOL, (tag, regret)
    return idx, oacq


def test_c1_full_size_ucb_vs_oracle_on_all_candidates(ta):
    cfg, X, y, Xc, ls = _inputs("c1")
    sur, model = _model(ta, cfg, X, y, ls)
    f = _acq(ta, cfg, model, y)
    assert f.get_name() == "-LCB"
    full = f(Xc)
    assert full.shape == (cfg["M"],) and np.all(np.isfinite(full))
    rng = np.random.RandomState(11)
    bi, bv = _common_properties(ta, cfg, sur, model, f, X, y, Xc, full, rng)
    om = _oracle(cfg, X, y, ls)
    assert model.get_log_likelihood() == pytest.approx(om.lml, rel=1e-9)
    omu, osg, oacq = _oracle_acq(cfg, om, Xc, y)                      # ALL 65 536 candidates
    mu, sg = model.predict(Xc, return_std_dev=True)
    _check_f64(cfg, om, mu, sg, full, omu, osg, oacq, "c1_f64")
    assert bi == int(np.argmax(oacq))


def test_c2_full_size_ei_vs_oracle_on_all_candidates(ta):
    cfg, X, y, Xc, ls = _inputs("c2")
    sur, model = _model(ta, cfg, X, y, ls)
    assert model.get_hyper_param_names()[1:3] == ["k1__k2__length_scale_0", "k1__k2__length_scale_1"]   # ARD
    f = _acq(ta, cfg, model, y)
    full = f(Xc)
    assert full.shape == (cfg["M"],) and np.all(np.isfinite(full)) and np.all(full >= 0)
    rng = np.random.RandomState(12)
    bi, bv = _common_properties(ta, cfg, sur, model, f, X, y, Xc, full, rng)
    om = _oracle(cfg, X, y, ls)
    assert model.get_log_likelihood() == pytest.approx(om.lml, rel=1e-9)
    omu, osg, oacq = _oracle_acq(cfg, om, Xc, y)                      # ALL 131 072 candidates
    mu, sg = model.predict(Xc, return_std_dev=True)
    _check_f64(cfg, om, mu, sg, full, omu, osg, oacq, "c2_f64")
    assert oacq[bi] >= oacq.max() * (1 - 1e-9)


def _split_dtypes_at_full_size(ta, cfg, X, y, Xc, ls, om, idx, omu, osg, oacq, tag):
    """the opt-in split-operand sweeps ('f32h2': two scaled fp16 planes, 'f32x3': three bf16 planes)
    at the configuration's full size: the f32 bounds against the oracle on the same >= 16 k
    candidates, shard invariance bit for bit, arg-max of the vector, regret judged by the oracle"""
    M = Xc.shape[0]
    for dt in ("f32h2", "f32x3"):
        surx, modelx = _model(ta, cfg, X, y, ls, dtype=dt)
        fx = _acq(ta, cfg, modelx, y)
        fullx = fx(Xc)
        assert fullx.shape == (M,) and np.all(np.isfinite(fullx))
        bix, bvx = fx.maximise(Xc)
        assert bix == int(np.argmax(fullx)) and bvx == fullx[bix]
        h = (M // 3) | 1
        np.testing.assert_array_equal(np.concatenate([fx(Xc[:h]), fx(Xc[h:])]), fullx)
        mux, sgx = modelx.predict(Xc[idx], return_std_dev=True)
        _check_f32(cfg, om, mux, sgx, omu, osg, "%s_%s" % (tag, dt))
        _, _, o_at = _oracle_acq(cfg, om, Xc[bix:bix + 1], y)
        best = max(float(oacq.max()), float(o_at[0]))
        regret = (best - float(o_at[0])) / max(abs(best), 1e-300)
        _measured["%s_%s_regret" % (tag, dt)] = dict(regret=regret, choice=int(bix))
        assert regret < REGRET_TOL, (tag, dt, regret)


def test_c3_full_size_f32_tolerance_and_regret(ta):
    cfg, X, y, Xc, ls = _inputs("c3")
    sur, model = _model(ta, cfg, X, y, ls)                            # f32 sweep
    f = _acq(ta, cfg, model, y)
    full = f(Xc)
    assert full.shape == (cfg["M"],) and np.all(np.isfinite(full)) and np.all(full >= 0)
    rng = np.random.RandomState(13)
    bi, bv = _common_properties(ta, cfg, sur, model, f, X, y, Xc, full, rng)
    om = _oracle(cfg, X, y, ls)
    assert model.get_log_likelihood() == pytest.approx(om.lml, rel=1e-9)     # the fit is f64
    # the oracle on ALL 262 144 candidates (about 16 s on the box's host cores): the f32 sweep inside its
    # bounds everywhere, the f64 sweep of the same model to the north_star tolerance everywhere, and the
    # arg-max regret exact (the oracle's own maximum over the whole batch, not over a proposal set)
    omu, osg, oacq = _oracle_acq(cfg, om, Xc, y, chunk=16384)
    mu, sg = model.predict(Xc, return_std_dev=True)
    _check_f32(cfg, om, mu, sg, omu, osg, "c3_f32")
    best = float(oacq.max())
    regret = (best - float(oacq[bi])) / max(abs(best), 1e-300)
    _measured["c3_regret"] = dict(regret=regret, oracle_best=best, oracle_at_choice=float(oacq[bi]), checked=int(len(oacq)))
    assert regret < REGRET_TOL, regret
    # round 4: the value REPORTED for the chosen point is formed once more in float64 (CandidateSweep, for f32
    # sweeps): it meets the fp64 bar where the sweep's own figure carries the f32 cross-kernel's rounding
    b = ta.Bounds([("x%d" % d, 0.0, 1.0) for d in range(cfg["D"])])
    x, info = ta.CandidateSweep(num_random=cfg["M"], gen_random=lambda n, lb: Xc)(b, f)
    np.testing.assert_array_equal(x[0], Xc[bi])
    err_sweep = abs(info["max_acq_sweep"] - float(oacq[bi])) / abs(float(oacq[bi]))
    err_f64 = abs(info["max_acq"] - float(oacq[bi])) / abs(float(oacq[bi]))
    _measured["c3_reported_value"] = dict(sweep_f32=info["max_acq_sweep"], refined_f64=info["max_acq"], oracle=float(oacq[bi]),
                                          rel_err_sweep=err_sweep, rel_err_refined=err_f64)
    assert info["max_acq_sweep"] == bv and err_f64 < RTOL and err_f64 <= err_sweep
    sur64, model64 = _model(ta, cfg, X, y, ls, dtype="f64")
    f64 = _acq(ta, cfg, model64, y)
    full64 = f64(Xc)
    mu64, sg64 = model64.predict(Xc, return_std_dev=True)
    _check_f64(cfg, om, mu64, sg64, full64, omu, osg, oacq, "c3_f64")
    assert oacq[f64.maximise(Xc)[0]] >= best * (1 - 1e-9)
    # the opt-in split-operand sweeps on the 2 048 best + 14 336 random candidates
    idx = np.unique(np.concatenate([np.argsort(-oacq, kind="stable")[:2048], rng.choice(cfg["M"], 14336, replace=False)]))
    _split_dtypes_at_full_size(ta, cfg, X, y, Xc, ls, om, idx, omu[idx], osg[idx], oacq[idx], "c3")


def test_c4_full_size_one_gpu_shard_and_oracle(ta):
    cfg, X, y, Xc, ls = _inputs("c4")
    M = cfg["M"]
    assert Xc.shape == (1048576, 64)
    sur, model = _model(ta, cfg, X, y, ls)                            # f32 sweep
    f = _acq(ta, cfg, model, y)
    assert f.get_name() == "PI"
    full = f(Xc)                                                      # the whole batch on one GPU
    assert full.shape == (M,) and np.all((full >= 0) & (full <= 1))
    bi, bv = f.maximise(Xc)
    assert bi == int(np.argmax(full)) and bv == full[bi]
    # the 8-GPU share: contiguous shards of M / 8 = 131 072 give the same values and, reduced with
    # the lowest-global-index rule, the same winner
    share = M // 8
    from turbo_amd.distributed import reduce_winners
    vals, idxs = [], []
    for r in (0, 3, 7):
        part = f(Xc[r * share:(r + 1) * share])
        np.testing.assert_array_equal(part, full[r * share:(r + 1) * share])
    for r in range(8):
        i, v = f.maximise(Xc[r * share:(r + 1) * share])
        vals.append(v)
        idxs.append(r * share + i)
    w = reduce_winners(vals, idxs)
    assert (idxs[w], vals[w]) == (bi, bv)
    rng = np.random.RandomState(14)
    perm = rng.permutation(4096)
    np.testing.assert_array_equal(f(Xc[:4096][perm]), full[:4096][perm])
    # oracle: fit, then >= 16 384 candidates (the best 2 048 by the f64 sweep + 16 384 random)
    om = _oracle(cfg, X, y, ls)
    assert model.get_log_likelihood() == pytest.approx(om.lml, rel=1e-8)
    sur64, model64 = _model(ta, cfg, X, y, ls, dtype="f64")
    full64 = _acq(ta, cfg, model64, y)(Xc)
    idx, oacq = _regret(cfg, om, y, Xc, bi, full64, 2048, 16384, rng, "c4_regret")
    assert len(idx) >= 16384
    omu, osg = o.predict(om, Xc[idx], True, chunk=4096)
    mu, sg = model.predict(Xc[idx], return_std_dev=True)
    _check_f32(cfg, om, mu, sg, omu, osg, "c4_f32")
    mu64, sg64 = model64.predict(Xc[idx], return_std_dev=True)
    _check_f64(cfg, om, mu64, sg64, full64[idx], omu, osg, oacq, "c4_f64")
    # PI values of the f32 sweep against the oracle's (Phi is 1-Lipschitz / sigma in its argument)
    assert np.max(np.abs(full[idx] - oacq)) < 5e-3
    _split_dtypes_at_full_size(ta, cfg, X, y, Xc, ls, om, idx, omu, osg, oacq, "c4")


def test_rccl_winner_exchange_on_one_rank():
    """bench.py's multi-GPU step over RCCL with a process group of one rank (tests/rccl_world1_check.py,
    in a process of its own): the device-packed record, the all-gather on the GPU, the reduce"""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MASTER_PORT", None)      # the child picks a free port
    out = subprocess.run([sys.executable, os.path.join(here, "rccl_world1_check.py")], env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0 and "rccl world-1 exchange ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
