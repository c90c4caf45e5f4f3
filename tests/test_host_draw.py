"""CPU: the reference's HOST candidate draw (turbo/modules/naive_selectors.py:39-46 -- one np.random.uniform per
parameter from NumPy's GLOBAL RNG, hstacked) continued by the library (tgp_mt19937_uniform_columns): the numbers and the
RNG state afterwards must be NumPy's own, bit for bit, whatever the batch shape, the bounds and the position inside
the generator's 624-word block -- a seeded run of the reference's Optimiser must not be able to tell the difference."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import turbo_amd as ta                       # noqa: E402
from turbo_amd import _lib                   # noqa: E402
from turbo_amd.naive_selectors import random_selector   # noqa: E402


def numpy_draw(M, lo, hi):
    """the reference's loop, verbatim in behaviour: a column per parameter, hstacked"""
    return np.hstack([np.random.uniform(a, b, size=(M, 1)) for a, b in zip(lo, hi)])


class Bounds:
    def __init__(self, lo, hi):
        self.ordered = [("p%d" % i, a, b) for i, (a, b) in enumerate(zip(lo, hi))]


@pytest.mark.parametrize("M,D,burn", [(1, 1, 0), (5, 3, 0), (311, 1, 7), (312, 2, 1), (313, 2, 622), (1000, 7, 623),
                                      (4099, 16, 5), (20000, 33, 100), (70001, 5, 311)])
def test_the_library_continues_numpys_global_stream_bit_for_bit(M, D, burn):
    rng = np.random.RandomState(M + D)
    lo = rng.uniform(-5, 5, D)
    hi = lo + rng.uniform(0.1, 20, D)
    if D > 2:
        hi[1] = lo[1]                # an empty range: every draw is lo
        lo[2], hi[2] = hi[2], lo[2]  # high < low: NumPy just computes low + (high - low) * u
    np.random.seed(1234 + M)
    np.random.randint(0, 10, size=burn)          # an arbitrary position inside the 624-word block (32-bit draws)
    np.random.standard_normal(3)                 # ... and a cached Gaussian in the state, which must survive
    want = numpy_draw(M, lo, hi)
    after_want = (np.random.uniform(size=5), np.random.standard_normal(2), np.random.randint(0, 1 << 30, size=3))
    np.random.seed(1234 + M)
    np.random.randint(0, 10, size=burn)
    np.random.standard_normal(3)
    got = _lib.numpy_global_uniform_columns(M, lo, hi)
    assert got is not None, "the library has the entry: the fast path must be taken"
    after_got = (np.random.uniform(size=5), np.random.standard_normal(2), np.random.randint(0, 1 << 30, size=3))
    assert got.shape == want.shape and got.dtype == np.float64 and got.flags.c_contiguous
    assert np.array_equal(got, want)
    for a, b in zip(after_want, after_got):
        assert np.array_equal(a, b), "NumPy's global RNG is not where its own calls would have left it"


def test_random_selector_is_the_reference_draw_on_both_paths():
    lo = [-5.0, 0.0, 1e-3]
    hi = [10.0, 15.0, 2e-3]
    for M in (7, 20000):
        np.random.seed(99)
        want = numpy_draw(M, lo, hi)
        tail = np.random.uniform(size=3)
        for fast_min in (0, 1 << 60):            # always the library / always NumPy's loop
            sel = random_selector()
            sel.FAST_DRAW_MIN = fast_min
            np.random.seed(99)
            got = sel(M, Bounds(lo, hi))
            assert np.array_equal(got, want) and np.array_equal(np.random.uniform(size=3), tail), (M, fast_min)


def test_what_the_library_cannot_promise_is_left_to_numpy():
    st = np.random.get_state()
    # a range that is not finite: NumPy raises OverflowError -- the wrapper must not draw anything
    assert _lib.numpy_global_uniform_columns(10, [0.0], [np.inf]) is None
    assert _lib.numpy_global_uniform_columns(10, [-1e308], [1e308]) is None
    assert _lib.numpy_global_uniform_columns(0, [0.0], [1.0]) is None
    st2 = np.random.get_state()
    assert st[2] == st2[2] and np.array_equal(st[1], st2[1]), "a refused draw must leave the global RNG untouched"
    sel = random_selector()
    sel.FAST_DRAW_MIN = 0
    with pytest.raises(OverflowError):
        sel(10, Bounds([0.0], [np.inf]))         # NumPy's own error, from NumPy's own loop


def test_the_entry_refuses_bad_arguments_without_touching_anything():
    lib = _lib.load()
    import ctypes
    key = np.arange(624, dtype=np.uint32)
    out = np.full((4, 2), 7.0)
    lo = np.zeros(2)
    hi = np.ones(2)
    for pos in (-1, 625):
        p = ctypes.c_int32(pos)
        rc = lib.tgp_mt19937_uniform_columns(key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(p), 4, 2, _lib._ptr(lo), _lib._ptr(hi), _lib._ptr(out))
        assert rc == _lib.BAD_ARG and p.value == pos and np.all(out == 7.0) and np.array_equal(key, np.arange(624, dtype=np.uint32))
    p = ctypes.c_int32(0)
    assert lib.tgp_mt19937_uniform_columns(None, ctypes.byref(p), 4, 2, _lib._ptr(lo), _lib._ptr(hi), _lib._ptr(out)) == _lib.BAD_ARG
    assert lib.tgp_mt19937_uniform_columns(key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(p), 0, 2, _lib._ptr(lo), _lib._ptr(hi), _lib._ptr(out)) == _lib.BAD_ARG


def test_the_host_only_library_has_the_entry_too():
    """a machine without ROCm (turbo_amd/_lib.py falls back to libturbogp_host.so) draws the same batch"""
    import ctypes
    if not os.path.exists(_lib.HOST_LIB_PATH):
        pytest.skip("libturbogp_host.so not built")
    host = ctypes.CDLL(_lib.HOST_LIB_PATH)
    np.random.seed(5)
    st = np.random.get_state()
    want = numpy_draw(1000, [0.0, -2.0], [1.0, 3.0])
    key = np.array(st[1], dtype=np.uint32)
    pos = ctypes.c_int32(int(st[2]))
    out = np.empty((1000, 2))
    lo, hi = np.array([0.0, -2.0]), np.array([1.0, 3.0])
    host.tgp_mt19937_uniform_columns.restype = ctypes.c_int
    rc = host.tgp_mt19937_uniform_columns(key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos), ctypes.c_int64(1000), ctypes.c_int64(2),
                                          lo.ctypes.data_as(ctypes.c_void_p), hi.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0 and np.array_equal(out, want)
    after = np.random.get_state()
    assert int(pos.value) == int(after[2]) and np.array_equal(key, after[1])


def test_known_answer_of_the_generator_itself():
    """MT19937 seeded by Matsumoto & Nishimura's init_genrand(5489), the reference implementation's default seed: the
    first outputs are 3499211612, 581869302, ... and the first 53-bit double 0.8147236863931789 (the published
    test vector of mt19937ar.c; also the first number MATLAB's default generator prints) -- held here without NumPy in
    the loop: the state is built by hand, the expected values are constants."""
    import ctypes
    key = np.empty(624, dtype=np.uint32)
    s = 5489
    for i in range(624):
        key[i] = s & 0xFFFFFFFF
        s = (1812433253 * (s ^ (s >> 30)) + i + 1) & 0xFFFFFFFF
    pos = ctypes.c_int32(624)
    out = np.empty((3, 1))
    lo, hi = np.array([0.0]), np.array([1.0])
    lib = _lib.load()
    rc = lib.tgp_mt19937_uniform_columns(key.ctypes.data_as(ctypes.c_void_p), ctypes.byref(pos), 3, 1, _lib._ptr(lo), _lib._ptr(hi), _lib._ptr(out))
    assert rc == _lib.OK and pos.value == 6
    words = [3499211612, 581869302, 3890346734, 3586334585, 545404204, 4161255391]
    want = [((words[2 * i] >> 5) * 67108864.0 + (words[2 * i + 1] >> 6)) / 9007199254740992.0 for i in range(3)]
    assert want[0] == 0.8147236863931789
    assert out[:, 0].tolist() == want


def test_a_draw_from_another_thread_waits_instead_of_being_lost():
    """the library holds NumPy's own generator lock from the read of the state to its write-back: numbers drawn by
    another thread meanwhile are neither repeated nor lost -- together the two threads consume the stream exactly once"""
    import threading
    M, D = 400000, 4
    lo, hi = np.zeros(D), np.ones(D)
    np.random.seed(8)
    ref = np.random.uniform(size=M * D + 4000)              # the stream, in order
    np.random.seed(8)
    got_small = []
    started = threading.Event()

    def small_draws():
        started.wait()
        for _ in range(40):
            got_small.append(np.random.uniform(size=100))
    t = threading.Thread(target=small_draws)
    t.start()
    started.set()
    big = _lib.numpy_global_uniform_columns(M, lo, hi)
    t.join()
    small = np.concatenate(got_small)
    rest = np.random.uniform(size=10)
    # every number of the stream's first M D + 4000 went to exactly one consumer, in order within each consumer
    used = np.concatenate([big.T.reshape(-1), small])       # (column c of `big` = draws [c M, (c + 1) M) of ITS part of the stream)
    assert np.array_equal(np.sort(used), np.sort(ref))
    assert np.array_equal(rest, np.random.RandomState(8).uniform(size=M * D + 4000 + 10)[-10:])
