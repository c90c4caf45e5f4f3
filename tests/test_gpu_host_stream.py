"""GPU: the reference's HOST candidate draw (turbo/modules/naive_selectors.py:39-46: NumPy's global RNG, a column per
parameter, hstacked) finished on the GPU -- tgp_set_candidates_mt19937 continues NumPy's MT19937 stream inside the
library and forms the doubles in a kernel.  The resident batch must be NumPy's array bit for bit, np.random must end
where NumPy's own calls would leave it, and a trial through the plugin classes must choose the point it chooses when
the batch is drawn by NumPy on the host."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def ta():
    import turbo_amd
    return turbo_amd


def numpy_draw(M, lo, hi):
    return np.hstack([np.random.uniform(a, b, size=(M, 1)) for a, b in zip(lo, hi)])


def _fitted(ta, N, D, dtype, seed):
    rng = np.random.RandomState(seed)
    X = rng.uniform(0, 1, (N, D))
    y = np.sin(3 * X.sum(1)) + 0.1 * rng.normal(size=N)
    gp = ta.NativeGP(0, dtype)
    gp.fit(X, y, "matern52", 1.0, 0.7, 1e-3, 1e-10, True)
    return gp, X, y


@pytest.mark.parametrize("M,D,burn,dtype", [(1, 1, 0, "f64"), (63, 3, 5, "f64"), (64, 32, 623, "f64"), (1000, 33, 1, "f32"),
                                            (50001, 7, 311, "f64"), (262144, 8, 17, "f32")])
def test_the_resident_batch_is_numpys_array_bit_for_bit(ta, M, D, burn, dtype):
    gp, _, _ = _fitted(ta, 40, D, dtype, D)
    rng = np.random.RandomState(M + D)
    lo = rng.uniform(-5, 5, D)
    hi = lo + rng.uniform(0.1, 20, D)      # (ranges whose product with u rounds differently under a fused multiply-add)
    if D > 2:
        hi[1] = lo[1]
        lo[2], hi[2] = hi[2], lo[2]
    np.random.seed(77 + M)
    np.random.randint(0, 10, size=burn)
    np.random.standard_normal(3)
    want = numpy_draw(M, lo, hi)
    after_want = (np.random.uniform(size=5), np.random.standard_normal(2))
    np.random.seed(77 + M)
    np.random.randint(0, 10, size=burn)
    np.random.standard_normal(3)
    assert gp.set_candidates_numpy_stream(M, lo, hi) is True
    after_got = (np.random.uniform(size=5), np.random.standard_normal(2))
    got = gp.read_candidates()
    assert got.shape == (M, D)
    assert np.array_equal(got, want), "max |diff| %g" % np.max(np.abs(got - want))
    for a, b in zip(after_want, after_got):
        assert np.array_equal(a, b)
    # ... and the sweep over it is the sweep over NumPy's array
    r1 = gp.sweep(ta._lib.ACQ_UCB, -1.0, 0.0, 2.0, want_acq=True)
    gp.set_candidates(want)
    r2 = gp.sweep(ta._lib.ACQ_UCB, -1.0, 0.0, 2.0, want_acq=True)
    assert r1["best_idx"] == r2["best_idx"] and r1["best_val"] == r2["best_val"] and np.array_equal(r1["acq"], r2["acq"])


def test_a_refused_draw_changes_nothing(ta):
    gp, _, _ = _fitted(ta, 30, 2, "f64", 1)
    np.random.seed(3)
    st = np.random.get_state()
    assert gp.set_candidates_numpy_stream(100, [0.0, 0.0], [1.0, np.inf]) is False       # NumPy raises there: left to NumPy
    assert gp.set_candidates_numpy_stream(100, [0.0], [1.0]) is False                    # not the model's D
    st2 = np.random.get_state()
    assert st[2] == st2[2] and np.array_equal(st[1], st2[1])
    host = ta.NativeGP(ta._lib.DEVICE_HOST, "f64")
    rng = np.random.RandomState(0)
    X = rng.uniform(0, 1, (10, 2))
    host.fit(X, X.sum(1), "rbf", 1.0, 0.5, 1e-3, 1e-10, True)
    assert host.set_candidates_numpy_stream(100, [0.0, 0.0], [1.0, 1.0]) is False        # a host handle: no GPU to finish it on
    # the entry itself: a bad position is refused, key and pos untouched
    import ctypes
    key = np.arange(624, dtype=np.uint32)
    pos = ctypes.c_int32(700)
    lo, hi = np.zeros(2), np.ones(2)
    rc = gp.lib.tgp_set_candidates_mt19937(gp._h, key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos), 10, 0, 10, ta._lib._ptr(lo), ta._lib._ptr(hi))
    assert rc == ta._lib.BAD_ARG and pos.value == 700 and np.array_equal(key, np.arange(624, dtype=np.uint32))
    pos = ctypes.c_int32(3)
    for total, first, rows in ((10, 5, 6), (10, -1, 3), (10, 0, 0)):      # rows outside the batch
        rc = gp.lib.tgp_set_candidates_mt19937(gp._h, key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos), total, first, rows,
                                               ta._lib._ptr(lo), ta._lib._ptr(hi))
        assert rc == ta._lib.BAD_ARG and pos.value == 3 and np.array_equal(key, np.arange(624, dtype=np.uint32))


@pytest.mark.parametrize("M,D,world", [(1000, 3, 2), (50001, 7, 8), (5, 2, 8)])
def test_shards_of_one_batch(ta, M, D, world):
    """what a rank of a sharded job keeps: rows [offset, offset + m_local) of the batch NumPy would have drawn for the
    whole job, with np.random behind the WHOLE batch afterwards -- ranks started with the same seed then sweep disjoint
    shards of the batch a single GPU sweeps (turbo_amd/distributed.py shard_plan)"""
    from turbo_amd.distributed import shard_plan
    gp, _, _ = _fitted(ta, 40, D, "f64", D)
    rng = np.random.RandomState(M)
    lo = rng.uniform(-5, 5, D)
    hi = lo + rng.uniform(0.1, 20, D)
    np.random.seed(11)
    np.random.randint(0, 10, size=13)
    whole = numpy_draw(M, lo, hi)
    after_want = np.random.uniform(size=5)
    seen = 0
    for rank in range(world):
        m_local, offset, _ = shard_plan(M, world, rank)
        np.random.seed(11)
        np.random.randint(0, 10, size=13)
        if m_local == 0:
            assert gp.set_candidates_numpy_stream(M, lo, hi, first=offset, count=0) is False     # nothing to keep: the caller passes over the batch itself
            continue
        assert gp.set_candidates_numpy_stream(M, lo, hi, first=offset, count=m_local) is True
        assert np.array_equal(np.random.uniform(size=5), after_want)
        got = gp.read_candidates()
        assert got.shape == (m_local, D) and np.array_equal(got, whole[offset:offset + m_local])
        seen += m_local
    assert seen == M


@pytest.mark.parametrize("grad_restarts,start_from_best", [(0, 0), (4, 3), (3, 0)])
def test_a_trial_through_the_plugin_classes_chooses_what_the_host_draw_chooses(ta, grad_restarts, start_from_best):
    """CandidateSweep with the default random_selector: the batch finished on the GPU (default for large batches) against
    NumPy's own loop + upload (STREAM_DRAW_MIN out of reach): the same point, the same value, the same RNG afterwards --
    also when the gradient stage starts from the best candidates (their rows then come back from the GPU) and when its
    starts are drawn behind the batch."""
    D, N, M = 5, 120, 60000
    rng = np.random.RandomState(5)
    X = rng.uniform(0, 1, (N, D))
    y = np.sin(3 * X.sum(1)) + 0.3 * ((X - 0.4) ** 2).sum(1) + 0.05 * rng.normal(size=N)
    b = ta.Bounds([("x%d" % d, -0.25 * d, 1.0 + 0.5 * d) for d in range(D)])
    out = {}
    for name, stream_min in (("stream", 1), ("host", 1 << 62)):
        sur = ta.HipGPSurrogate(model_params=dict(kernel=ta.GPKernel("matern52", 1.0, 0.8, 1e-3), optimizer=None, normalize_y=True,
                                                  alpha=1e-10), training_iterations=1, dtype="f64")
        model, _ = sur.construct_model(0, X, y)
        f = ta.EI(0.01).construct_function(0, model, "min", float(y.min()))[0]
        aux = ta.CandidateSweep(num_random=M, grad_restarts=grad_restarts, start_from_best=start_from_best)
        aux.STREAM_DRAW_MIN = stream_min
        np.random.seed(2024)
        x, info = aux(b, f)
        out[name] = (x, info["max_acq"], np.random.uniform(size=4))
    assert np.array_equal(out["stream"][0], out["host"][0])
    assert out["stream"][1] == out["host"][1]
    assert np.array_equal(out["stream"][2], out["host"][2])
