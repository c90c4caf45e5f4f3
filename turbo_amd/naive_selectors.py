"""Candidate / pre-phase selectors (turbo/modules/naive_selectors.py).

``random_selector`` (:39-46) and ``LHS_selector`` (:58-83) with the reference's call contract
``selector(num_points, latent_bounds) -> (num_points, D)``.  ``random_selector`` draws exactly like the
reference (the golden Branin trace depends on that draw order); ``LHS_selector``'s host design consumes the
global NumPy RNG in the reference's order too -- the jitter of all cells first, then one permutation per
dimension -- so a seeded run gives the reference's pre-phase points whether or not the reference is installed
(tests/test_interop.py holds the two to the bit where it is).  With ``device_seed`` the Latin hypercube design
is drawn ON the GPU instead (``tgp_lhs_design``: a keyed permutation of the strata per dimension and
the jitter from the Philox stream of that seed), which needs no fitted model -- the reference uses
this selector for the pre-phase trials, before any surrogate exists.  The candidate batch of the
acquisition sweep can use the same design without ever leaving the GPU:
``CandidateSweep(device_rng_seed=..., device_design='lhs')``.
"""
import numpy as np


class random_selector:
    """points uniform-random in the latent space: one column per parameter from the global
    NumPy RNG, hstacked (turbo/modules/naive_selectors.py:39-46).

    Large batches (``num_points x D >= FAST_DRAW_MIN``) are drawn by the library's own continuation of NumPy's global
    MT19937 stream (``tgp_mt19937_uniform_columns``): the same numbers, the same RNG state afterwards -- bit for bit
    what the loop below returns (tests/test_host_draw.py) -- in a fraction of the time (C3's 262 144 x 32: 66 ms in
    NumPy on the GPU box's host, twice the GPU step the batch feeds)."""

    FAST_DRAW_MIN = 1 << 15      # elements; below, the interpreter's own loop costs microseconds

    def __call__(self, num_points, latent_bounds):
        ordered = latent_bounds.ordered
        if num_points * len(ordered) >= self.FAST_DRAW_MIN:
            from . import _lib
            try:
                lows = [float(b[1]) for b in ordered]
                highs = [float(b[2]) for b in ordered]
            except (TypeError, ValueError):
                lows = None
            out = _lib.numpy_global_uniform_columns(num_points, lows, highs) if lows is not None else None
            if out is not None:
                return out
        cols = []
        for name, pmin, pmax in ordered:
            cols.append(np.random.uniform(pmin, pmax, size=(num_points, 1)))
        return np.hstack(cols)


class LHS_selector:
    """Latin Hypercube sampling selector with the reference's contract
    (turbo/modules/naive_selectors.py:58-83): a design of ``num_total`` points is fixed at the first call and
    handed out in consecutive slices.

    * ``device_seed`` given: the design comes from the GPU (``tgp_lhs_design``).
    * otherwise: ONE host construction, everywhere -- ``rand(n, D)`` for the position inside each cell, stratum k of
      every dimension filled in row k, then the rows of each dimension re-ordered by a permutation of its own,
      dimension by dimension.  That is the order in which the reference draws from the global RNG
      (turbo/modules/naive_selectors.py:75-78), so ``np.random.seed(s)`` gives the same points here and there.
      (Round 4 delegated to the reference's class when importable and built another design otherwise: the same seed
      then gave different pre-phase points on a machine without the reference.)
    """

    def __init__(self, num_total, device_seed=None, device=0):
        """
        Args:
            num_total: length of the sequence (number of strata per dimension)
            device_seed: None (default) = a host design from the global NumPy RNG (see above).  An
                integer = the design comes from the GPU (Philox stream ``device_seed``); slices are
                then computed on demand, any slice equals the same rows of the whole design.
            device: HIP device index for the device-side design
        """
        self.num_total = num_total
        self.sequence = None
        self.index = 0  # index into the sequence
        self.device_seed = device_seed
        self.device = device
        self._ctx = None

    def __call__(self, num_points, latent_bounds):
        assert self.index + num_points <= self.num_total, 'LHS sequence exhausted!'
        lo = np.array([b[1] for b in latent_bounds.ordered], dtype=np.float64)
        hi = np.array([b[2] for b in latent_bounds.ordered], dtype=np.float64)
        if self.device_seed is not None:
            from . import _lib
            if self._ctx is None:
                self._ctx = _lib.NativeGP(self.device, 'f64')
            samples = self._ctx.lhs_design(self.device_seed, self.index, num_points, self.num_total, lo, hi)
        else:
            if self.sequence is None:
                n, dims = self.num_total, len(lo)
                within = np.random.rand(n, dims)                       # drawn first ...
                design = lo + (hi - lo) * (np.arange(n)[:, None] + within) / n
                for d in range(dims):                                  # ... then one permutation per dimension
                    design[:, d] = design[np.random.permutation(n), d]
                self.sequence = design
            samples = self.sequence[self.index:self.index + num_points, :]
        self.index += num_points
        return samples

    def __getstate__(self):
        d = dict(self.__dict__)
        d['_ctx'] = None        # the GPU context is not picklable
        return d
