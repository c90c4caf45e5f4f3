"""ctypes binding of libturbogp.so (the C-ABI in include/turbogp.h).

The optimisation path has no CPU fallback: if the library is missing or cannot be loaded this module
raises ``TurboGPLibraryError`` on first use, and every native class in this package is unusable.
The one thing that works without a GPU is the RELOAD path (a pickled model queried where no HIP device
is visible): ``NativeGP(DEVICE_HOST)`` -- a handle of the library's host backend -- and, where not even
the ROCm runtime is installed, the host-only build ``libturbogp_host.so`` (``load()`` says which).
"""
import ctypes
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

# Hardware queues.  The HIP runtime deals a process's streams round-robin onto GPU_MAX_HW_QUEUES hardware queues
# (default 4) and two streams on one queue run one after the other: with the library's three shared streams per
# device plus the private streams of a threaded hyper-parameter fit (three per factory) a second live factory
# doubled the time of a fit (N = 500: 11.6 -> 19.3 ms; 11.8 with eight queues).  The runtime
# reads the variable ONCE, when it initialises (the first HIP call of the process: here tgp_create, or torch's own
# first CUDA call if that came earlier -- then this default is too late and the process keeps four queues unless
# the variable was exported before Python started).  An explicit setting in the environment always wins.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
# TGP_LIBRARY points at another build of the same library (kernel experiments)
LIB_PATH = os.environ.get("TGP_LIBRARY") or os.path.join(_HERE, "csrc", "libturbogp.so")

OK, NOT_PD, BAD_ARG, HIP_ERROR, NOT_FITTED, NO_MEMORY, NO_DEVICE = 0, 1, 2, 3, 4, 5, 6
DEVICE_HOST = -1     # tgp_create(TGP_DEVICE_HOST): the reload path without a GPU (include/turbogp.h)
F64, F32, F32X3, F32H2 = 0, 1, 2, 3
KERNELS = {"rbf": 0, "matern12": 1, "matern32": 2, "matern52": 3}
ACQ_NONE, ACQ_UCB, ACQ_PI, ACQ_EI, ACQ_SIGMA = 0, 1, 2, 3, 4
BUF_K, BUF_L, BUF_LINV, BUF_ALPHA = 0, 1, 2, 3

# every symbol include/turbogp.h declares
SYMBOLS = (
    "tgp_create", "tgp_destroy", "tgp_last_error", "tgp_version", "tgp_fit", "tgp_fit_grad", "tgp_fit_optimise", "tgp_fit_lbfgsb", "tgp_set_private_stream",
    "tgp_set_overlap", "tgp_tuning", "tgp_mt19937_uniform_columns", "tgp_set_candidates_mt19937", "tgp_stream_status", "tgp_workers_acquire", "tgp_workers_release",
    "tgp_fit_append", "tgp_export_state", "tgp_import_state", "tgp_export_factor_dev", "tgp_import_factor_dev", "tgp_debug_read",
    "tgp_set_candidates", "tgp_set_candidates_dev", "tgp_gen_candidates", "tgp_gen_candidates_lhs", "tgp_lhs_design",
    "tgp_read_candidates", "tgp_get_candidate",
    "tgp_sweep", "tgp_sweep_topk", "tgp_set_winner_out", "tgp_winner_wait", "tgp_acq_grad", "tgp_acq_refine", "tgp_acq_lbfgsb",
    "tgp_evaluate", "tgp_predict_batch", "tgp_predict", "tgp_profile_enable", "tgp_profile_read", "tgp_profile_reset",
    "tgp_sweep_geometry", "tgp_last_timings",
    "tgp_multi_create", "tgp_multi_destroy", "tgp_multi_last_error", "tgp_multi_size", "tgp_multi_handle",
    "tgp_multi_fit", "tgp_multi_set_candidates", "tgp_multi_gen_candidates", "tgp_multi_sweep",
)


class TurboGPLibraryError(RuntimeError):
    """libturbogp.so is missing / not loadable / reported a HIP failure."""


class NoDeviceError(TurboGPLibraryError):
    """tgp_create found no HIP device (TGP_NO_DEVICE)."""


_lib = None


class Factor(ctypes.Structure):
    """``tgp_factor`` (include/turbogp.h): what a sweep needs of a fit, as device pointers"""
    _fields_ = [("N", ctypes.c_int64), ("D", ctypes.c_int64), ("Np", ctypes.c_int64), ("Dp", ctypes.c_int64),
                ("fit_gen", ctypes.c_int64), ("kernel", ctypes.c_int32), ("normalize_y", ctypes.c_int32),
                ("small_path", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("constant", ctypes.c_double), ("noise", ctypes.c_double), ("jitter", ctypes.c_double),
                ("y_mean", ctypes.c_double), ("y_std", ctypes.c_double), ("lml", ctypes.c_double), ("sumlog", ctypes.c_double),
                ("Xs", ctypes.c_void_p), ("ls", ctypes.c_void_p), ("alpha", ctypes.c_void_p), ("Linv", ctypes.c_void_p)]

_dp = ctypes.POINTER(ctypes.c_double)
_i64p = ctypes.POINTER(ctypes.c_int64)
_vp = ctypes.c_void_p


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.

    PyTorch's ROCm wheels bundle their own ``libamdhip64.so`` (SONAME ``libamdhip64.so.7``, the
    same as /opt/rocm's).  If ``torch`` is imported first, libturbogp.so's dependency resolves
    to that already-loaded copy and all is well; if libturbogp.so comes first it pulls in
    /opt/rocm's copy and a later ``import torch`` loads a SECOND runtime, which then finds "No
    HIP GPUs" -- and device pointers of one runtime are unknown to the other
    (``tgp_set_candidates_dev`` checks them).  So when PyTorch is installed, bind to its copy
    up front, without importing torch.  ``TGP_HIP_RUNTIME=system`` keeps /opt/rocm's."""
    if os.environ.get("TGP_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass   # libturbogp.so then binds to the system runtime as linked


HOST_LIB_PATH = os.path.join(_HERE, "csrc", "libturbogp_host.so")
HOST_ONLY = False     # True once load() had to settle for libturbogp_host.so (no ROCm runtime on this machine)
LOAD_ERROR = None     # why libturbogp.so itself could not be loaded, when HOST_ONLY (the dlopen message)


def _argtypes():
    c = ctypes
    fit = [_vp, _dp, c.c_int64, c.c_int64, _dp, c.c_int, c.c_double, _dp, c.c_int64, c.c_double, c.c_double,
           c.c_int, _dp, _dp, _dp]
    return {
        "tgp_last_error": [_vp],
        "tgp_create": [c.c_int, c.c_int, c.POINTER(_vp)],
        "tgp_destroy": [_vp],
        "tgp_fit": fit,
        "tgp_fit_grad": fit + [_dp],
        "tgp_fit_append": fit + [c.POINTER(c.c_int)],
        "tgp_fit_optimise": [_vp, _dp, c.c_int64, c.c_int64, _dp, c.c_int, _dp, c.c_int64, c.c_int64, _dp, _dp,
                             c.c_double, c.c_int, c.c_int64, _dp, _dp, _i64p, _i64p],
        "tgp_fit_lbfgsb": [_vp, _dp, c.c_int64, c.c_int64, _dp, c.c_int, _dp, c.c_int64, c.c_int64, _dp, _dp,
                           c.c_double, c.c_int, c.c_int64, _dp, _dp, _i64p, _i64p],
        "tgp_export_state": [_vp, _vp, c.c_int64, _i64p],
        "tgp_import_state": [_vp, _vp, c.c_int64, _dp],
        "tgp_export_factor_dev": [_vp, c.POINTER(Factor)],
        "tgp_import_factor_dev": [_vp, c.POINTER(Factor), c.c_int64, c.c_int64],
        "tgp_debug_read": [_vp, c.c_int, _dp],
        "tgp_set_candidates": [_vp, _dp, c.c_int64],
        "tgp_set_candidates_dev": [_vp, _vp, c.c_int64],
        "tgp_gen_candidates": [_vp, c.c_uint64, c.c_uint64, c.c_int64, _dp, _dp],
        "tgp_gen_candidates_lhs": [_vp, c.c_uint64, c.c_uint64, c.c_int64, c.c_uint64, _dp, _dp],
        "tgp_lhs_design": [_vp, c.c_uint64, c.c_uint64, c.c_int64, c.c_uint64, c.c_int64, _dp, _dp, _dp],
        "tgp_read_candidates": [_vp, c.c_int64, c.c_int64, _dp],
        "tgp_get_candidate": [_vp, c.c_int64, _dp],
        "tgp_sweep": [_vp, c.c_int, c.c_double, c.c_double, c.c_double, _dp, _dp, _dp, _dp, _i64p, _i64p],
        "tgp_sweep_topk": [_vp, c.c_int, c.c_double, c.c_double, c.c_double, c.c_int64, _dp, _i64p, _i64p],
        "tgp_acq_refine": [_vp, _dp, c.c_int64, _dp, _dp, c.c_int, c.c_double, c.c_double, c.c_double,
                           c.c_int64, _dp, _dp, _i64p, _i64p],
        "tgp_acq_lbfgsb": [_vp, _dp, c.c_int64, _dp, _dp, c.c_int, c.c_double, c.c_double, c.c_double,
                           c.c_int64, _dp, _dp, _i64p, _i64p],
        "tgp_set_winner_out": [_vp, _vp, c.c_int64],
        "tgp_winner_wait": [_vp, _vp],
        "tgp_stream_status": [_vp, c.POINTER(c.c_int)],
        "tgp_acq_grad": [_vp, _dp, c.c_int64, c.c_int, c.c_double, c.c_double, c.c_double, _dp, _dp],
        "tgp_evaluate": [_vp, _dp, c.c_int64, c.c_int, c.c_double, c.c_double, c.c_double, _dp, _dp, _dp,
                         _dp, _i64p, _i64p],
        "tgp_predict_batch": [_vp, c.c_int64, _i64p, c.c_int64, c.POINTER(_vp), c.POINTER(_vp), c.c_int, _dp, _dp,
                              _dp, _dp, c.c_int, _dp, c.c_int64, _dp, _dp, _dp, _i64p],
        "tgp_predict": [_vp, _dp, c.c_int64, _dp, _dp],
        "tgp_profile_enable": [_vp, c.c_int],
        "tgp_set_private_stream": [_vp, c.c_int],
        "tgp_set_overlap": [_vp, c.c_int],
        "tgp_workers_acquire": [_vp, c.c_int, c.POINTER(_vp)],
        "tgp_workers_release": [_vp],
        "tgp_tuning": [c.c_char_p, c.c_int64],
        "tgp_mt19937_uniform_columns": [_vp, c.POINTER(c.c_int32), c.c_int64, c.c_int64, _dp, _dp, _dp],
        "tgp_set_candidates_mt19937": [_vp, _vp, c.POINTER(c.c_int32), c.c_int64, c.c_int64, c.c_int64, _dp, _dp],
        "tgp_profile_read": [_vp, _i64p, _dp, _i64p, _dp, _dp, _dp],
        "tgp_profile_reset": [_vp],
        "tgp_sweep_geometry": [_vp, _i64p, _i64p],
        "tgp_last_timings": [_vp, _dp, c.c_int64],
        "tgp_multi_last_error": [_vp],
        "tgp_multi_create": [c.c_int, c.POINTER(c.c_int), c.c_int, c.POINTER(_vp)],
        "tgp_multi_destroy": [_vp],
        "tgp_multi_size": [_vp],
        "tgp_multi_handle": [_vp, c.c_int, c.POINTER(_vp)],
        "tgp_multi_fit": [_vp] + fit[1:],
        "tgp_multi_set_candidates": [_vp, _dp, c.c_int64],
        "tgp_multi_gen_candidates": [_vp, c.c_uint64, c.c_int64, _dp, _dp],
        "tgp_multi_sweep": [_vp, c.c_int, c.c_double, c.c_double, c.c_double, _dp, _i64p, _dp, _dp],
    }


class _Unavailable:
    """stands in for a GPU-only entry of the host-only library: calling it says what is missing"""

    def __init__(self, name):
        self.name = name

    def __call__(self, *a):
        raise TurboGPLibraryError("%s needs the GPU build of libturbogp.so; this process loaded the host-only "
                                  "library (%s), which serves reloaded models only" % (self.name, HOST_LIB_PATH))


def load():
    """Load the library once and declare argument types.  Raises loudly when absent.

    Where libturbogp.so cannot be LOADED -- a machine without the ROCm runtime it links, e.g. the laptop
    that plots a recorder -- the host-only build libturbogp_host.so (same C-ABI names, TGP_DEVICE_HOST
    handles only) is taken instead and ``HOST_ONLY`` is set: reloaded models predict, everything
    else raises."""
    global _lib, HOST_ONLY, LOAD_ERROR
    if _lib is not None:
        return _lib
    if os.environ.get("TGP_LIBRARY") and not os.path.exists(LIB_PATH):
        # an explicit path that does not exist is a mistake, never a reason to settle for the host-only build
        raise TurboGPLibraryError("TGP_LIBRARY=%s does not exist" % LIB_PATH)
    if not os.path.exists(LIB_PATH) and not os.path.exists(HOST_LIB_PATH):
        raise TurboGPLibraryError(
            "libturbogp.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C turbo_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    lib = None
    err = None
    if os.path.exists(LIB_PATH):
        _share_hip_runtime_with_torch()
        try:
            lib = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            err = e
    if lib is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise TurboGPLibraryError("cannot load %s: %s" % (LIB_PATH, err)) from err
        try:
            lib = ctypes.CDLL(HOST_LIB_PATH)
        except OSError as e:
            raise TurboGPLibraryError("cannot load %s (%s) nor %s (%s)" % (LIB_PATH, err, HOST_LIB_PATH, e)) from e
    lib.tgp_version.restype = ctypes.c_char_p
    HOST_ONLY = b"host-only" in lib.tgp_version()
    if HOST_ONLY:
        LOAD_ERROR = ("%s: %s" % (LIB_PATH, err)) if err is not None else "%s is not there" % LIB_PATH
    for name, args in _argtypes().items():
        if not hasattr(lib, name):
            if not HOST_ONLY:
                raise TurboGPLibraryError("%s does not export %s" % (LIB_PATH, name))
            setattr(lib, name, _Unavailable(name))
            continue
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = (ctypes.c_char_p if name in ("tgp_last_error", "tgp_multi_last_error")
                      else ctypes.c_int64 if name == "tgp_tuning" else ctypes.c_int)
    _lib = lib
    return lib


class _Workers:
    def __init__(self, gp, n):
        self.gp, self.n = gp, n

    def __enter__(self):
        arr = (_vp * self.n)()
        self.gp._check(self.gp.lib.tgp_workers_acquire(self.gp._h, self.n, arr))
        # the pool's mutex is this thread's from here: whatever goes wrong before the block is entered (a failed
        # allocation, a KeyboardInterrupt) must give it back, or every later hyper-parameter fit on the device blocks
        try:
            self.views = [NativeGP._borrowed(self.gp.lib, _vp(arr[i]), self.gp.device) for i in range(self.n)]
        except BaseException:
            self.gp.lib.tgp_workers_release(self.gp._h)
            raise
        return self.views

    def __exit__(self, *exc):
        # the handles go back to the pool: the views handed out must not outlive the block (the library may destroy the
        # workers with the device's last ordinary handle, and another thread's fit may be using them by then)
        for w in getattr(self, "views", []):
            w._h = None
        self.views = []
        rc = self.gp.lib.tgp_workers_release(self.gp._h)
        if rc != OK and exc[0] is None:
            self.gp._check(rc)
        return False


class _GlobalRngLock:
    """NumPy's own lock around its global generator -- every draw of ``np.random`` takes it -- held while the library reads
    the state, continues the stream and writes the state back: another thread's draw then waits, exactly as it would
    behind one of NumPy's own long draws, instead of landing between the read and the write-back and being lost."""

    def __enter__(self):
        try:
            self.lock = np.random.mtrand._rand._bit_generator.lock
            self.lock.acquire()
        except Exception:
            self.lock = None
        return self

    def __exit__(self, *exc):
        if self.lock is not None:
            self.lock.release()
        return False


def numpy_global_uniform_columns(num_points, lows, highs):
    """``np.hstack([np.random.uniform(lo, hi, size=(num_points, 1)) for lo, hi in zip(lows, highs)])`` -- the reference's
    ``random_selector`` (turbo/modules/naive_selectors.py:39-46) -- computed by ``tgp_mt19937_uniform_columns``: NumPy's
    GLOBAL legacy RNG is read (``np.random.get_state``), its MT19937 stream continued in C++ and the state written back,
    so the numbers AND every later draw from ``np.random`` are the ones NumPy itself would have produced, bit for bit
    (tests/test_host_draw.py).  Returns None where that cannot be promised (another bit generator behind the global
    RNG, bounds whose range is not finite, a library without the entry): the caller then asks NumPy."""
    lo = np.asarray(lows, dtype=np.float64).reshape(-1)
    hi = np.asarray(highs, dtype=np.float64).reshape(-1)
    if num_points < 1 or lo.size < 1 or lo.shape != hi.shape:
        return None
    with np.errstate(over="ignore", invalid="ignore"):
        if not np.all(np.isfinite(hi - lo)):
            return None   # (NumPy raises for a range that is not finite: let it)
    try:
        lib = load()
        entry = lib.tgp_mt19937_uniform_columns
    except Exception:
        return None
    if isinstance(entry, _Unavailable):
        return None
    out = np.empty((int(num_points), lo.size), dtype=np.float64)
    lo, hi = _f64c(lo), _f64c(hi)
    with _GlobalRngLock():
        st = np.random.get_state()
        if not isinstance(st, tuple) or st[0] != "MT19937" or len(st) != 5:
            return None
        key = np.array(st[1], dtype=np.uint32, order="C", copy=True)
        pos = ctypes.c_int32(int(st[2]))
        rc = entry(key.ctypes.data_as(_vp), ctypes.byref(pos), int(num_points), lo.size, _ptr(lo), _ptr(hi), _ptr(out))
        if rc != OK:
            return None       # (nothing was written back: NumPy's state is untouched)
        np.random.set_state((st[0], key, int(pos.value), st[3], st[4]))
    return out


def tuning():
    """every TGP_* switch of the library with the value in force in this process (``tgp_tuning``):
    {name: (value, description)}"""
    lib = load()
    n = lib.tgp_tuning(None, 0)
    buf = ctypes.create_string_buffer(int(n))
    lib.tgp_tuning(buf, n)
    out = {}
    for line in buf.value.decode().splitlines():
        kv, _, doc = line.partition("\t# ")
        k, _, v = kv.partition("=")
        out[k] = (v, doc)
    return out


def _ptr(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def _f64c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class NativeGP:
    """One GPU context (tgp_handle).  Thin, stateful, not thread-safe per handle.
    ``device=DEVICE_HOST`` makes a host context instead (``self.host``): fit / predict / acquisition of
    a reloaded model without a GPU, nothing else."""

    def __init__(self, device=0, dtype="f64"):
        self._h = None
        self.lib = load()
        assert dtype in ("f64", "f32", "f32x3", "f32h2"), "dtype must be 'f64', 'f32', 'f32x3' or 'f32h2'"
        self.dtype = dtype
        self.device = int(device)
        self.host = self.device == DEVICE_HOST
        h = _vp()
        rc = self.lib.tgp_create(self.device, {"f64": F64, "f32": F32, "f32x3": F32X3, "f32h2": F32H2}[dtype], ctypes.byref(h))
        if rc == NO_DEVICE:
            raise NoDeviceError(self.lib.tgp_last_error(None).decode())
        if rc != OK:
            raise TurboGPLibraryError(self.lib.tgp_last_error(None).decode())
        self._h = h
        self._cand_keepalive = None
        self._winner_keepalive = None
        self.gen_key = None      # what the resident batch IS when the GPU generated it: (kind, seed, first, M, total, lo, hi)
        if not self.host:
            self._warn_if_streams_serialised()

    _stream_warned = set()       # devices the warning below was given for (once per process and device)

    def stream_status(self):
        """``tgp_stream_status``: {'background_overlaps', 'third_overlaps'} (1 / 0 / -1 = not probed) and
        'gpu_max_hw_queues' as the environment has it (0: unset)"""
        v = (ctypes.c_int * 3)()
        self._check(self.lib.tgp_stream_status(self._h, v))
        return dict(background_overlaps=int(v[0]), third_overlaps=int(v[1]), gpu_max_hw_queues=int(v[2]))

    def _warn_if_streams_serialised(self):
        if self.device in NativeGP._stream_warned or not hasattr(self.lib, "tgp_stream_status"):
            return
        try:
            st = self.stream_status()
        except Exception:       # (a library without the entry: TGP_LIBRARY pointing at an older build)
            return
        if st["background_overlaps"] == 0 or st["third_overlaps"] == 0:
            NativeGP._stream_warned.add(self.device)
            import warnings
            what = []
            if st["background_overlaps"] == 0:
                what.append("the fit's background stream runs on the main stream's hardware queue: the inverse factor "
                            "follows the Cholesky instead of running beside it (large fits take about twice as long)")
            if st["third_overlaps"] == 0:
                what.append("there is no third stream: tgp_set_overlap / CandidateSweep(prefetch_next=True) is a no-op")
            warnings.warn("turbo_amd: the library's streams on device %d probed as SERIALISED (%s).  The HIP runtime deals "
                          "streams onto GPU_MAX_HW_QUEUES hardware queues (4 by default) and reads the variable once, at its "
                          "first call in the process: turbo_amd sets 8 at import, which is too late when another library "
                          "(torch.cuda) initialised HIP first -- GPU_MAX_HW_QUEUES is %s here.  Export GPU_MAX_HW_QUEUES=8 "
                          "before starting Python, or import turbo_amd and create its first context before touching "
                          "torch.cuda." % (self.device, "; ".join(what),
                                          st["gpu_max_hw_queues"] if st["gpu_max_hw_queues"] else "unset"), RuntimeWarning)

    def close(self):
        if getattr(self, "_h", None) is not None:
            if getattr(self, "_owned", True):
                self.lib.tgp_destroy(self._h)
            self._h = None

    @classmethod
    def _borrowed(cls, lib, handle, device):
        """a view of a handle the library owns (a pooled worker): every call works, close() does not destroy"""
        w = cls.__new__(cls)
        w._h, w.lib, w.dtype, w.device, w.host = handle, lib, "f64", int(device), False
        w._cand_keepalive = w._winner_keepalive = None
        w.gen_key = None
        w._owned = False
        return w

    def workers(self, n):
        """``tgp_workers_acquire``: a context manager yielding n worker handles of this handle's device (each on a
        private stream), the device's pool locked for this thread until the block ends"""
        return _Workers(self, int(n))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == OK:
            return
        msg = self.lib.tgp_last_error(self._h).decode()
        if rc == NOT_PD:
            # same exception type (and advice) as sklearn/gaussian_process/_gpr.py:348-358
            raise np.linalg.LinAlgError(
                "The kernel is not returning a positive definite matrix. Try gradually "
                "increasing the 'alpha' parameter of the surrogate. (%s)" % msg)
        if rc == BAD_ARG:
            raise ValueError(msg)
        if rc == NOT_FITTED:
            raise RuntimeError(msg)
        if rc == NO_MEMORY:
            raise MemoryError(msg)
        raise TurboGPLibraryError(msg)

    def fit(self, X, y, kind, constant, length_scale, noise, jitter, normalize_y, append=False):
        """full fit, or (append=True) a one-row extension of the resident factor when X is the
        resident training set plus one row; ``self.appended`` tells which path ran"""
        X = _f64c(X)
        y = _f64c(y).reshape(-1)
        assert X.ndim == 2 and X.shape[0] == y.shape[0], "X must be (N, D) and y (N,)"
        ls = _f64c(np.atleast_1d(length_scale))
        lml, ym, ys = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        args = (self._h, _ptr(X), X.shape[0], X.shape[1], _ptr(y), KERNELS[kind], float(constant),
                _ptr(ls), ls.shape[0], float(noise), float(jitter), 1 if normalize_y else 0,
                ctypes.byref(lml), ctypes.byref(ym), ctypes.byref(ys))
        self.appended = False
        if append:
            flag = ctypes.c_int(0)
            self._check(self.lib.tgp_fit_append(*args, ctypes.byref(flag)))
            self.appended = bool(flag.value)
        else:
            self._check(self.lib.tgp_fit(*args))
        self.N, self.D = X.shape
        return lml.value, ym.value, ys.value

    def fit_grad(self, X, y, kind, constant, length_scale, noise, jitter, normalize_y):
        """fit + d LML / d log(theta): returns (lml, grad [c, ls..., noise])"""
        X = _f64c(X)
        y = _f64c(y).reshape(-1)
        assert X.ndim == 2 and X.shape[0] == y.shape[0], "X must be (N, D) and y (N,)"
        ls = _f64c(np.atleast_1d(length_scale))
        lml, ym, ys = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        grad = np.zeros(2 + ls.shape[0])
        self._check(self.lib.tgp_fit_grad(
            self._h, _ptr(X), X.shape[0], X.shape[1], _ptr(y), KERNELS[kind], float(constant),
            _ptr(ls), ls.shape[0], float(noise), float(jitter), 1 if normalize_y else 0,
            ctypes.byref(lml), ctypes.byref(ym), ctypes.byref(ys), _ptr(grad)))
        self.N, self.D = X.shape
        return lml.value, grad

    def set_private_stream(self, on=True):
        """submit this handle's work to a stream of its own (handles of a device share one by default),
        so calls on several handles from several threads overlap on the GPU"""
        self._check(self.lib.tgp_set_private_stream(self._h, 1 if on else 0))

    def set_overlap(self, mode=2):
        """``tgp_set_overlap``: the next fits start the sweep of the RESIDENT candidate batch inside themselves
        (1: candidate scaling + the first cross-kernel, 2: + the contraction's early row tiles; 0: off).
        Results are bit-identical to the serial schedule."""
        self._check(self.lib.tgp_set_overlap(self._h, int(mode)))

    def fit_optimise(self, X, y, kind, theta0, n_ls, log_bounds, jitter, normalize_y, max_iter=500, lbfgsb=False):
        """the hyper-parameter fit inside the library: every row of theta0 (S, 2 + n_ls) = log(constant, length
        scale(s), noise) is optimised inside log_bounds (2 + n_ls, 2); returns (theta (S, P), -lml (S,), status (S,),
        evaluations).  ``lbfgsb=True`` = ``tgp_fit_lbfgsb``: SciPy's L-BFGS-B restated in the library, the iterates
        scikit-learn's fit walks, at every size; False = ``tgp_fit_optimise``: one launch with a projected L-BFGS for
        N <= 128, ``tgp_fit_lbfgsb`` above."""
        X = _f64c(X)
        y = _f64c(y).reshape(-1)
        theta0 = _f64c(np.atleast_2d(theta0))
        S, P = theta0.shape
        assert P == 2 + n_ls, "theta0 must be (S, 2 + n_ls)"
        lb = _f64c(np.asarray(log_bounds, dtype=np.float64).reshape(P, 2))
        lo, hi = _f64c(lb[:, 0]), _f64c(lb[:, 1])
        theta = np.empty((S, P))
        f = np.empty(S)
        st = np.empty(S, dtype=np.int64)
        ev = ctypes.c_int64(0)
        entry = self.lib.tgp_fit_lbfgsb if lbfgsb else self.lib.tgp_fit_optimise
        self._check(entry(
            self._h, _ptr(X), X.shape[0], X.shape[1], _ptr(y), KERNELS[kind], _ptr(theta0), S, int(n_ls),
            _ptr(lo), _ptr(hi), float(jitter), 1 if normalize_y else 0, int(max_iter), _ptr(theta), _ptr(f),
            st.ctypes.data_as(_i64p), ctypes.byref(ev)))
        return theta, f, st, ev.value

    def export_state(self):
        """bytes that define the fitted model (theta, X, y) -- see tgp_export_state"""
        need = ctypes.c_int64(0)
        self._check(self.lib.tgp_export_state(self._h, None, 0, ctypes.byref(need)))
        buf = ctypes.create_string_buffer(need.value)
        self._check(self.lib.tgp_export_state(self._h, ctypes.cast(buf, _vp), need.value, ctypes.byref(need)))
        return buf.raw

    def import_state(self, blob):
        """rebuild the model of `export_state` on this handle (runs the fit); returns the LML"""
        blob = bytes(blob)
        lml = ctypes.c_double(0.0)
        buf = ctypes.create_string_buffer(blob, len(blob))
        self._check(self.lib.tgp_import_state(self._h, ctypes.cast(buf, _vp), len(blob), ctypes.byref(lml)))
        n, d = np.frombuffer(blob, dtype=np.int64, count=2, offset=8)
        self.N, self.D = int(n), int(d)
        return lml.value

    def export_factor(self):
        """``tgp_export_factor_dev``: a ``Factor`` of device pointers into THIS handle's buffers (valid until its next
        fit / import / close) -- what another handle, or another rank after a broadcast, needs to sweep with this fit"""
        f = Factor()
        self._check(self.lib.tgp_export_factor_dev(self._h, ctypes.byref(f)))
        return f

    def import_factor(self, factor, row0=0, rows=None):
        """``tgp_import_factor_dev``: receive rows [row0, row0 + rows) of a factor (all of it by default); rows arrive in
        order, the handle is fitted when the last one is in.  Returns True once the factor is complete."""
        rows = int(factor.Np - row0) if rows is None else int(rows)
        self._check(self.lib.tgp_import_factor_dev(self._h, ctypes.byref(factor), int(row0), rows))
        self.N, self.D = int(factor.N), int(factor.D)
        return row0 + rows == factor.Np

    def debug_read(self, which):
        N = self.N
        out = np.empty(N if which == BUF_ALPHA else (N, N), dtype=np.float64)
        self._check(self.lib.tgp_debug_read(self._h, which, _ptr(out)))
        return out

    def set_candidates(self, Xc):
        Xc = _f64c(Xc)
        assert Xc.ndim == 2 and Xc.shape[1] == self.D, "candidates must be (M, %d)" % self.D
        self._check(self.lib.tgp_set_candidates(self._h, _ptr(Xc), Xc.shape[0]))
        self.M = Xc.shape[0]
        self._cand_keepalive = None
        self.gen_key = None

    def set_candidates_numpy_stream(self, M, lo, hi, first=0, count=None):
        """make resident the batch ``np.hstack([np.random.uniform(l, h, size=(M, 1)) for l, h in zip(lo, hi)])`` -- the
        reference's ``random_selector`` draw (turbo/modules/naive_selectors.py:39-46) -- WITHOUT forming it on the host:
        NumPy's global MT19937 stream is continued in the library, the GPU forms the doubles
        (``tgp_set_candidates_mt19937``), and ``np.random`` is left where its own calls would have left it.  Returns
        False -- nothing drawn, nothing changed -- where that cannot be promised (a host handle, another bit generator
        behind the global RNG, a range that is not finite): the caller draws with NumPy then.

        ``first`` / ``count``: keep only rows [first, first + count) of that M-row batch resident -- a rank's shard of ONE
        batch -- while ``np.random`` still ends behind the whole batch (the other rows' numbers are passed over)."""
        lo, hi = _f64c(lo).reshape(-1), _f64c(hi).reshape(-1)
        count = int(M) - int(first) if count is None else int(count)
        if getattr(self, "host", False) or HOST_ONLY or lo.shape != (self.D,) or hi.shape != (self.D,) or int(M) < 1:
            return False
        if first < 0 or count < 1 or first + count > int(M):
            return False
        with np.errstate(over="ignore", invalid="ignore"):
            if not np.all(np.isfinite(hi - lo)):
                return False
        with _GlobalRngLock():
            st = np.random.get_state()
            if not isinstance(st, tuple) or st[0] != "MT19937" or len(st) != 5:
                return False
            key = np.array(st[1], dtype=np.uint32, order="C", copy=True)
            pos = ctypes.c_int32(int(st[2]))
            self._check(self.lib.tgp_set_candidates_mt19937(self._h, key.ctypes.data_as(_vp), ctypes.byref(pos), int(M), int(first), count,
                                                            _ptr(lo), _ptr(hi)))
            np.random.set_state((st[0], key, int(pos.value), st[3], st[4]))
        self.M = count
        self._cand_keepalive = None
        self.gen_key = None
        return True

    def set_candidates_dev(self, dev_ptr, M, keepalive=None):
        self._check(self.lib.tgp_set_candidates_dev(self._h, _vp(int(dev_ptr)), int(M)))
        self.M = int(M)
        self._cand_keepalive = keepalive
        self.gen_key = None

    def gen_candidates(self, seed, first_candidate, M, lo, hi):
        """M uniform candidates in [lo, hi) drawn on the GPU (Philox-4x32-10 stream `seed`)"""
        lo, hi = _f64c(lo).reshape(-1), _f64c(hi).reshape(-1)
        assert lo.shape == hi.shape == (self.D,), "bounds must have one entry per dimension"
        self._check(self.lib.tgp_gen_candidates(self._h, int(seed), int(first_candidate), int(M),
                                                _ptr(lo), _ptr(hi)))
        self.M = int(M)
        self._cand_keepalive = None
        self.gen_key = ("uniform", int(seed), int(first_candidate), int(M), None, lo.tobytes(), hi.tobytes())

    def gen_candidates_lhs(self, seed, first_sample, M, n_total, lo, hi):
        """rows first_sample .. + M of an n_total-point Latin hypercube design drawn on the GPU become
        the resident batch"""
        lo, hi = _f64c(lo).reshape(-1), _f64c(hi).reshape(-1)
        assert lo.shape == hi.shape == (self.D,), "bounds must have one entry per dimension"
        self._check(self.lib.tgp_gen_candidates_lhs(self._h, int(seed), int(first_sample), int(M),
                                                    int(n_total), _ptr(lo), _ptr(hi)))
        self.M = int(M)
        self._cand_keepalive = None
        self.gen_key = ("lhs", int(seed), int(first_sample), int(M), int(n_total), lo.tobytes(), hi.tobytes())

    def lhs_design(self, seed, first_sample, M, n_total, lo, hi):
        """(M, D) rows of an n_total-point Latin hypercube design, drawn on the GPU, as a host array
        (no fitted model needed)"""
        lo, hi = _f64c(lo).reshape(-1), _f64c(hi).reshape(-1)
        assert lo.shape == hi.shape and lo.ndim == 1
        out = np.empty((int(M), lo.shape[0]))
        self.gen_key = None      # the design overwrites the handle's own batch: a generated batch is no longer resident
        self._check(self.lib.tgp_lhs_design(self._h, int(seed), int(first_sample), int(M), int(n_total),
                                            lo.shape[0], _ptr(lo), _ptr(hi), _ptr(out)))
        return out

    def read_candidates(self, first=0, count=None):
        count = self.M - first if count is None else count
        out = np.empty((int(count), self.D))
        self._check(self.lib.tgp_read_candidates(self._h, int(first), int(count), _ptr(out)))
        return out

    def get_candidate(self, idx):
        out = np.empty(self.D, dtype=np.float64)
        self._check(self.lib.tgp_get_candidate(self._h, int(idx), _ptr(out)))
        return out

    def sweep(self, acq=ACQ_NONE, sf=1.0, incumbent=0.0, param=0.0, want_mu=False,
              want_sigma=False, want_acq=False):
        M = self.M
        mu = np.empty(M) if want_mu else None
        sg = np.empty(M) if want_sigma else None
        aq = np.empty(M) if want_acq else None
        bv, bi, nc = ctypes.c_double(float("nan")), ctypes.c_int64(-1), ctypes.c_int64(0)
        self._check(self.lib.tgp_sweep(self._h, acq, float(sf), float(incumbent), float(param),
                                       _ptr(mu), _ptr(sg), _ptr(aq), ctypes.byref(bv),
                                       ctypes.byref(bi), ctypes.byref(nc)))
        return dict(mu=mu, sigma=sg, acq=aq, best_val=bv.value, best_idx=bi.value,
                    n_clamped=nc.value, sweep_ms=self.profile_read()['last_sweep_ms'])

    def sweep_topk(self, k, acq, sf=1.0, incumbent=0.0, param=0.0):
        """the k best resident candidates: (indices (k,), values (k,)), best first"""
        vals = np.empty(int(k))
        idxs = np.empty(int(k), dtype=np.int64)
        nc = ctypes.c_int64(0)
        self._check(self.lib.tgp_sweep_topk(self._h, acq, float(sf), float(incumbent), float(param), int(k),
                                            _ptr(vals), idxs.ctypes.data_as(_i64p), ctypes.byref(nc)))
        return idxs, vals

    def acq_refine(self, X0, lo, hi, acq, sf=1.0, incumbent=0.0, param=0.0, max_iter=200, lbfgsb=False):
        """refine R start points together, maximising the acquisition: ``tgp_acq_refine`` (a projected L-BFGS on the
        GPU) or, ``lbfgsb=True``, ``tgp_acq_lbfgsb`` (L-BFGS-B itself in the library, SciPy's walk per restart, one
        batched evaluation per round); returns (x (R, D), values (R,), status (R,), evaluations)"""
        X0 = _f64c(np.atleast_2d(X0))
        assert X0.ndim == 2 and X0.shape[1] == self.D, "start points must be (R, %d)" % self.D
        lo, hi = _f64c(lo).reshape(-1), _f64c(hi).reshape(-1)
        assert lo.shape == hi.shape == (self.D,), "bounds must have one entry per dimension"
        R = X0.shape[0]
        x = np.empty((R, self.D))
        v = np.empty(R)
        st = np.empty(R, dtype=np.int64)
        its = ctypes.c_int64(0)
        entry = self.lib.tgp_acq_lbfgsb if lbfgsb else self.lib.tgp_acq_refine
        self._check(entry(self._h, _ptr(X0), R, _ptr(lo), _ptr(hi), acq, float(sf),
                          float(incumbent), float(param), int(max_iter), _ptr(x), _ptr(v),
                          st.ctypes.data_as(_i64p), ctypes.byref(its)))
        return x, v, st, its.value

    def set_winner_out(self, dev_ptr, global_offset=0, keepalive=None):
        """attach (or, with None, detach) the (D + 2,) float64 device record every sweep packs its
        winner [value, global index, row] into -- the input of the sharded arg-max's all-gather"""
        self._check(self.lib.tgp_set_winner_out(self._h, _vp(int(dev_ptr)) if dev_ptr else None,
                                                int(global_offset)))
        self._winner_keepalive = keepalive

    def winner_wait(self, stream=None):
        """``tgp_winner_wait``: make a stream of the caller (``torch.cuda.current_stream().cuda_stream`` -- an integer --
        or None for the default stream) wait for the winner record of the last sweep: the explicit ordering between the
        library's stream, which packs the record, and the stream RCCL reads it on"""
        self._check(self.lib.tgp_winner_wait(self._h, _vp(int(stream)) if stream else None))

    def evaluate(self, Xc, acq=ACQ_NONE, sf=1.0, incumbent=0.0, param=0.0, want_mu=False,
                 want_sigma=False, want_acq=False):
        """set_candidates + sweep in ONE call (tgp_evaluate); same result dict as ``sweep``"""
        Xc = _f64c(Xc)
        assert Xc.ndim == 2 and Xc.shape[1] == self.D, "candidates must be (M, %d)" % self.D
        M = Xc.shape[0]
        mu = np.empty(M) if want_mu else None
        sg = np.empty(M) if want_sigma else None
        aq = np.empty(M) if want_acq else None
        bv, bi, nc = ctypes.c_double(float("nan")), ctypes.c_int64(-1), ctypes.c_int64(0)
        self._check(self.lib.tgp_evaluate(self._h, _ptr(Xc), M, acq, float(sf), float(incumbent),
                                          float(param), _ptr(mu), _ptr(sg), _ptr(aq),
                                          ctypes.byref(bv), ctypes.byref(bi), ctypes.byref(nc)))
        self.M = M
        self._cand_keepalive = None
        self.gen_key = None
        return dict(mu=mu, sigma=sg, acq=aq, best_val=bv.value, best_idx=bi.value,
                    n_clamped=nc.value, sweep_ms=self.profile_read()['last_sweep_ms'])

    def predict_batch(self, models, Xc, want_sigma=True):
        """T stored models (N <= 256, same kernel kind / D / normalize_y) x one batch of points in one
        call.  ``models``: list of dicts with X (N, D), y (N,), kind, constant, length_scale, noise,
        jitter, normalize_y.  Returns (mu (T, M), sigma (T, M) or None, lml (T,), n_clamped)."""
        T = len(models)
        Xc = _f64c(np.atleast_2d(Xc))
        D = Xc.shape[1]
        kind, norm = models[0]['kind'], bool(models[0]['normalize_y'])
        Xs = [_f64c(m['X']) for m in models]
        ys = [_f64c(m['y']).reshape(-1) for m in models]
        for m, X, y in zip(models, Xs, ys):
            assert m['kind'] == kind and bool(m['normalize_y']) == norm, "one kernel kind / normalisation per batch"
            assert X.ndim == 2 and X.shape[1] == D and X.shape[0] == y.shape[0]
        Ns = np.array([X.shape[0] for X in Xs], dtype=np.int64)
        ls = np.vstack([np.broadcast_to(np.asarray(m['length_scale'], dtype=np.float64), (D,)) for m in models])
        ls = _f64c(ls)
        cs = _f64c([m['constant'] for m in models])
        ns = _f64c([m['noise'] for m in models])
        js = _f64c([m['jitter'] for m in models])
        xp = (_vp * T)(*[X.ctypes.data for X in Xs])
        yp = (_vp * T)(*[y.ctypes.data for y in ys])
        M = Xc.shape[0]
        mu = np.empty((T, M))
        sg = np.empty((T, M)) if want_sigma else None
        lml = np.empty(T)
        nc = ctypes.c_int64(0)
        self._check(self.lib.tgp_predict_batch(self._h, T, Ns.ctypes.data_as(_i64p), D, xp, yp, KERNELS[kind],
                                               _ptr(cs), _ptr(ls), _ptr(ns), _ptr(js), 1 if norm else 0,
                                               _ptr(Xc), M, _ptr(mu), _ptr(sg), _ptr(lml), ctypes.byref(nc)))
        return mu, sg, lml, nc.value

    def acq_grad(self, Xq, acq=ACQ_NONE, sf=1.0, incumbent=0.0, param=0.0):
        """acquisition value (m,) and gradient (m, D) at a small batch of points"""
        Xq = _f64c(np.atleast_2d(Xq))
        assert Xq.ndim == 2 and Xq.shape[1] == self.D, "points must be (m, %d)" % self.D
        val = np.empty(Xq.shape[0])
        grad = np.empty(Xq.shape)
        self._check(self.lib.tgp_acq_grad(self._h, _ptr(Xq), Xq.shape[0], acq, float(sf), float(incumbent),
                                          float(param), _ptr(val), _ptr(grad)))
        return val, grad

    def profile_enable(self, on=True):
        self._check(self.lib.tgp_profile_enable(self._h, 1 if on else 0))

    def profile_reset(self):
        self._check(self.lib.tgp_profile_reset(self._h))

    def profile_read(self):
        tl, kl = ctypes.c_int64(), ctypes.c_int64()
        tm, km, fm, sm = (ctypes.c_double() for _ in range(4))
        self._check(self.lib.tgp_profile_read(self._h, ctypes.byref(tl), ctypes.byref(tm),
                                              ctypes.byref(kl), ctypes.byref(km),
                                              ctypes.byref(fm), ctypes.byref(sm)))
        return dict(trmm_launches=tl.value, trmm_ms=tm.value, kstar_launches=kl.value,
                    kstar_ms=km.value, last_fit_ms=fm.value, last_sweep_ms=sm.value)

    def last_timings(self):
        """device times (ms) of the last calls: fit, sweep, and the three stages of the LML gradient"""
        v = np.zeros(12)
        self._check(self.lib.tgp_last_timings(self._h, _ptr(v), 12))
        return dict(fit_ms=v[0], sweep_ms=v[1], grad_kinv_ms=v[2], grad_pairwise_ms=v[3], grad_ard_ms=v[4],
                    trmm_flops=v[5], sweep_f64=int(v[6]), small_fit_phases_us=[float(x) for x in v[7:12]])

    def sweep_geometry(self):
        ch, npad = ctypes.c_int64(), ctypes.c_int64()
        self._check(self.lib.tgp_sweep_geometry(self._h, ctypes.byref(ch), ctypes.byref(npad)))
        return ch.value, npad.value
