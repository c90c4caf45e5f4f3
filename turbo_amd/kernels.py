"""Kernel specification for the native GP: ``c * k(r) [+ noise * delta]``.

The reference passes a scikit-learn kernel object in ``model_params['kernel']``
(turbo/modules/surrogates.py:231-243, :298).  ``GPKernel.from_any`` accepts either a
``GPKernel`` or such an object, recognised by duck typing on the class names so scikit-learn is
not imported here:  ``[ConstantKernel *] (RBF | Matern(nu in {0.5, 1.5, 2.5})) [+ WhiteKernel]``.
Hyper-parameter names follow scikit-learn's ``kernel_.hyperparameters`` naming so that
``get_hyper_param_names()`` lines up with the reference's SciKitGPSurrogate
(turbo/modules/surrogates.py:340-362).
"""
import math

import numpy as np

_NU_TO_KIND = {0.5: "matern12", 1.5: "matern32", 2.5: "matern52"}
KINDS = ("rbf", "matern12", "matern32", "matern52")


def _exp_like_numpy(t):
    """libm's exp (the bits the C++ side of the library computes) with numpy's answer outside its range"""
    try:
        return math.exp(t)
    except OverflowError:
        return float('inf')


class GPKernel:
    DEFAULT_BOUNDS = (1e-5, 1e5)   # scikit-learn's default for every hyper-parameter

    def __init__(self, kind="matern52", constant=1.0, length_scale=1.0, noise=None,
                 _names=None, _fixed=None, bounds=None):
        """
        Args:
            kind: 'rbf', 'matern12', 'matern32' or 'matern52'
            constant: ConstantKernel value c (> 0)
            length_scale: scalar (isotropic) or (D,) array (ARD)
            noise: WhiteKernel noise level, or None for no WhiteKernel term
            bounds: optional dict {'constant' | 'length_scale' | 'noise': (low, high) or 'fixed'}
                used by hyper-parameter optimisation (default (1e-5, 1e5) each)
        """
        assert kind in KINDS, "unknown kernel kind: {}".format(kind)
        self.kind = kind
        self.constant = float(constant)
        ls = np.asarray(length_scale, dtype=np.float64)
        self.length_scale = float(ls) if ls.ndim == 0 else ls.copy()
        self.noise = None if noise is None else float(noise)
        assert self.constant > 0, "constant must be > 0"
        assert np.all(np.asarray(self.length_scale) > 0), "length scales must be > 0"
        assert self.noise is None or self.noise >= 0, "noise must be >= 0"
        if _names is None:
            if self.noise is None:
                _names = {"constant": "k1__constant_value", "length_scale": "k2__length_scale"}
            else:
                _names = {"constant": "k1__k1__constant_value",
                          "length_scale": "k1__k2__length_scale", "noise": "k2__noise_level"}
        self._names = _names            # present keys only, insertion order = sklearn order
        self._fixed = set(_fixed or ())
        self._bounds = {}
        for key in self._names:
            b = (bounds or {}).get(key, self.DEFAULT_BOUNDS)
            if isinstance(b, str):
                assert b == 'fixed', "bounds must be (low, high) or 'fixed'"
                self._fixed.add(key)
                b = self.DEFAULT_BOUNDS
            self._bounds[key] = (float(b[0]), float(b[1]))

    def copy(self):
        """an independent copy (the surrogate must not change the kernel the caller handed over:
        turbo/modules/surrogates.py:299 deep-copies ``model_params``); ~20x cheaper than deepcopy"""
        k = GPKernel.__new__(GPKernel)
        k.kind = self.kind
        k.constant = self.constant
        k.length_scale = self.length_scale if np.ndim(self.length_scale) == 0 else np.array(self.length_scale, dtype=np.float64)
        k.noise = self.noise
        k._names = dict(self._names)
        k._fixed = set(self._fixed)
        k._bounds = dict(self._bounds)
        return k

    @property
    def noise_level(self):
        return 0.0 if self.noise is None else self.noise

    @property
    def anisotropic(self):
        return np.ndim(self.length_scale) > 0

    # ---- log-space parameter vector of the non-fixed hyper-parameters (sklearn's theta) ----
    def _free_keys(self):
        return [k for k in self._names if k not in self._fixed]

    @property
    def theta(self):
        return np.log(self.hyper_params()) if self._free_keys() else np.array([])

    @theta.setter
    def theta(self, theta):
        theta = np.asarray(theta, dtype=np.float64)
        # the C library's exp, entry by entry: numpy's vectorised exp differs from it in the last bit for about one
        # argument in twenty, and the library's own optimiser (csrc/host_lbfgsb.hpp, which calls exp from C++) is held
        # to walk the iterates SciPy walks on this objective -- to the evaluation, which only holds if both drivers
        # hand the GPU the same hyper-parameters bit for bit
        # (math.exp raises OverflowError beyond ~709.78 where np.exp returns inf: an unbounded theta -- optimizer='scipy'
        # or a user's optimiser callable -- may probe such values in a line search, and the fit must then take the
        # not-positive-definite / infinite-objective path, not crash)
        vals = np.array([_exp_like_numpy(t) for t in theta.reshape(-1)], dtype=np.float64)
        i = 0
        for key in self._free_keys():
            if key == "constant":
                self.constant = float(vals[i]); i += 1
            elif key == "noise":
                self.noise = float(vals[i]); i += 1
            else:
                if self.anisotropic:
                    n = len(self.length_scale)
                    self.length_scale = vals[i:i + n].copy(); i += n
                else:
                    self.length_scale = float(vals[i]); i += 1
        assert i == len(vals), "theta has the wrong length"

    @property
    def theta_bounds(self):
        """(n_free, 2) log-space bounds, one row per entry of theta (kernel_.bounds)"""
        rows = []
        for key in self._free_keys():
            lo, hi = np.log(self._bounds[key])
            n = len(self.length_scale) if (key == "length_scale" and self.anisotropic) else 1
            rows.extend([[lo, hi]] * n)
        return np.array(rows, dtype=np.float64).reshape(-1, 2)

    def select_gradient(self, grad):
        """pick the entries of the native gradient [c, ls..., noise] that belong to theta"""
        nl = len(self.length_scale) if self.anisotropic else 1
        parts = {"constant": grad[0:1], "length_scale": grad[1:1 + nl], "noise": grad[1 + nl:2 + nl]}
        sel = [parts[k] for k in self._free_keys()]
        return np.concatenate(sel) if sel else np.array([])

    def hyper_params(self):
        """values of the non-fixed hyper-parameters (surrogates.py:340-348)"""
        vals = []
        for key, _ in self._names.items():
            if key in self._fixed:
                continue
            v = {"constant": self.constant, "length_scale": self.length_scale,
                 "noise": self.noise}[key]
            vals.append(np.atleast_1d(np.asarray(v, dtype=np.float64)))
        return np.hstack(vals) if vals else np.array([])

    def hyper_param_names(self):
        """names, ARD length scales expanded to name_0.. (surrogates.py:350-362)"""
        names = []
        for key, name in self._names.items():
            if key == "length_scale" and self.anisotropic:
                names.extend("{}_{}".format(name, i) for i in range(len(self.length_scale)))
            else:
                names.append(name)
        return names

    def __repr__(self):
        return "GPKernel(kind={!r}, constant={!r}, length_scale={!r}, noise={!r})".format(
            self.kind, self.constant, self.length_scale, self.noise)

    # ---- scikit-learn kernel objects, by duck typing --------------------------------------
    @staticmethod
    def from_any(k):
        if isinstance(k, GPKernel):
            return k
        leaves = []
        _walk(k, "", leaves)
        const = stat = white = None
        for path, leaf in leaves:
            cls = type(leaf).__name__
            if cls == "ConstantKernel" and const is None:
                const = (path, leaf)
            elif cls in ("RBF", "Matern") and stat is None:
                stat = (path, leaf)
            elif cls == "WhiteKernel" and white is None:
                white = (path, leaf)
            else:
                raise ValueError("unsupported kernel structure: {!r}".format(k))
        if stat is None:
            raise ValueError("kernel must contain an RBF or Matern term: {!r}".format(k))
        _check_structure(k)
        path, leaf = stat
        if type(leaf).__name__ == "RBF":
            kind = "rbf"
        else:
            nu = float(leaf.nu)
            if nu == float("inf"):
                kind = "rbf"
            elif nu in _NU_TO_KIND:
                kind = _NU_TO_KIND[nu]
            else:
                raise ValueError("Matern nu={} is not supported (0.5, 1.5, 2.5, inf)".format(nu))
        names, fixed, bounds = {}, set(), {}

        def reg(key, pl, attr):
            names[key] = pl[0] + attr
            b = getattr(pl[1], attr + "_bounds", None)
            if isinstance(b, str):
                fixed.add(key)
            elif b is not None:
                b = np.asarray(b, dtype=np.float64).reshape(-1, 2)
                bounds[key] = (float(b[0, 0]), float(b[0, 1]))
        ordered = sorted([p for p in (("constant", const, "constant_value"),
                                      ("length_scale", stat, "length_scale"),
                                      ("noise", white, "noise_level")) if p[1] is not None],
                         key=lambda p: [l[0] for l in leaves].index(p[1][0]))
        for key, pl, attr in ordered:
            reg(key, pl, attr)
        return GPKernel(kind=kind,
                        constant=1.0 if const is None else float(const[1].constant_value),
                        length_scale=np.asarray(leaf.length_scale, dtype=np.float64),
                        noise=None if white is None else float(white[1].noise_level),
                        _names=names, _fixed=fixed, bounds=bounds)


def _walk(k, prefix, out):
    cls = type(k).__name__
    if cls in ("Sum", "Product"):
        _walk(k.k1, prefix + "k1__", out)
        _walk(k.k2, prefix + "k2__", out)
    else:
        out.append((prefix, k))


def _check_structure(k):
    """accept exactly  [Const *] Stat  [+ White]  (either operand order)"""
    cls = type(k).__name__

    def is_prod_or_stat(n):
        c = type(n).__name__
        if c in ("RBF", "Matern"):
            return True
        if c == "Product":
            pair = {type(n.k1).__name__, type(n.k2).__name__}
            return pair in ({"ConstantKernel", "RBF"}, {"ConstantKernel", "Matern"})
        return False
    if cls == "Sum":
        a, b = type(k.k1).__name__, type(k.k2).__name__
        ok = (b == "WhiteKernel" and is_prod_or_stat(k.k1)) or \
             (a == "WhiteKernel" and is_prod_or_stat(k.k2))
    else:
        ok = is_prod_or_stat(k)
    if not ok:
        raise ValueError("unsupported kernel structure (need [Constant *] RBF|Matern "
                         "[+ White]): {!r}".format(k))
