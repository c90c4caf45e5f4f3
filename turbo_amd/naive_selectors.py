"""Candidate / pre-phase selectors (turbo/modules/naive_selectors.py).

``random_selector`` (:39-46) and ``LHS_selector`` (:58-83) with the reference's call contract
``selector(num_points, latent_bounds) -> (num_points, D)``.  By default both draw exactly like the
reference, on the host from the global NumPy RNG.  With ``device_seed`` the Latin hypercube design
is drawn ON the GPU instead (``tgp_lhs_design``: a keyed permutation of the strata per dimension and
the jitter from the Philox stream of that seed), which needs no fitted model -- the reference uses
this selector for the pre-phase trials, before any surrogate exists.  The candidate batch of the
acquisition sweep can use the same design without ever leaving the GPU:
``CandidateSweep(device_rng_seed=..., device_design='lhs')``.
"""
import numpy as np


class random_selector:
    """points uniform-random in the latent space: one column per parameter from the global
    NumPy RNG, hstacked (turbo/modules/naive_selectors.py:39-46)"""

    def __call__(self, num_points, latent_bounds):
        cols = []
        for name, pmin, pmax in latent_bounds.ordered:
            cols.append(np.random.uniform(pmin, pmax, size=(num_points, 1)))
        return np.hstack(cols)


class LHS_selector:
    """Latin Hypercube sampling selector (turbo/modules/naive_selectors.py:58-83): a sequence of
    ``num_total`` points is fixed at the first call and handed out in consecutive slices."""

    def __init__(self, num_total, device_seed=None, device=0):
        """
        Args:
            num_total: length of the sequence (number of strata per dimension)
            device_seed: None (default) = the reference's host construction from the global NumPy
                RNG.  An integer = the design comes from the GPU (Philox stream ``device_seed``);
                slices are then computed on demand, any slice equals the same rows of the whole
                design.
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
        lower_bounds = np.array([b[1] for b in latent_bounds.ordered], dtype=np.float64)
        upper_bounds = np.array([b[2] for b in latent_bounds.ordered], dtype=np.float64)
        if self.device_seed is not None:
            from . import _lib
            if self._ctx is None:
                self._ctx = _lib.NativeGP(self.device, 'f64')
            samples = self._ctx.lhs_design(self.device_seed, self.index, num_points, self.num_total,
                                           lower_bounds, upper_bounds)
        else:
            if self.sequence is None:
                # first call, generate the sequence: fills points uniformly in each interval,
                # then shuffles each dimension (naive_selectors.py:66-78)
                n = self.num_total
                dims = len(latent_bounds.ordered)
                ranges = upper_bounds - lower_bounds
                self.sequence = lower_bounds + ranges * (np.arange(n).reshape(-1, 1) + np.random.rand(n, dims)) / n
                for i in range(dims):
                    self.sequence[:, i] = np.random.permutation(self.sequence[:, i])
            samples = self.sequence[self.index:self.index + num_points, :]
        self.index += num_points
        return samples

    def __getstate__(self):
        d = dict(self.__dict__)
        d['_ctx'] = None        # the GPU context is not picklable
        return d
