"""Acquisition-function plugins evaluated on the GPU, fused with the posterior sweep.

Mirror of the reference's factories and function instances
(turbo/modules/acquisition_functions.py): ``AcquisitionFunction`` :12-77, ``UCB`` :80-158,
``PI`` :163-247, ``EI`` :250-358 -- same constructor arguments, ``get_type()``,
``construct_function(trial_num, model, desired_extremum[, incumbent_cost])``, ``get_name()`` and
``__call__(X (M, D)) -> (M,)``.  With a native model (``HipGPSurrogate.ModelInstance``) mean, variance
and the acquisition value are produced by one pass of the HIP kernels.  A FOREIGN model -- any other
``Surrogate.ModelInstance`` of the reference, e.g. its ``SciKitGPSurrogate`` -- is served as the reference
serves it: ``model.predict(X, return_std_dev=True)`` and the formula in NumPy
(``_from_mu_sigma`` below; interoperability only -- a native model never takes that route, and the
GPU-only extras ``maximise_topk`` / ``refine`` / ``value_and_grad`` / ``maximise_generated`` refuse it).

Beyond the reference interface each instance has ``maximise(X) -> (index, value)``, which the
``CandidateSweep`` auxiliary optimiser uses to get the arg-max without copying M values back.
"""
from math import isinf

import numpy as np

from . import _lib


def _is_native(model):
    return hasattr(model, '_sweep')


def _require_native(model, what):
    if not _is_native(model):
        raise TypeError('{} needs a model built by HipGPSurrogate (got {!r}): it runs on the GPU only'
                        .format(what, type(model)))


def _from_mu_sigma(acq, sf, incumbent, param, mu, sigma):
    """UCB / PI / EI / sigma from a foreign model's posterior, float64 on the host
    (turbo/modules/acquisition_functions.py:147-158, :225-247, :336-358):
        UCB  sf mu + beta sigma                       (ACQ_SIGMA, i.e. beta = inf: sigma alone)
        PI   Phi(z),                z = (sf (mu - f+) - xi) / sigma
        EI   (sigma z) Phi(z) + sigma phi(z)          both 0 wherever sigma == 0"""
    from scipy.special import ndtr
    mu = np.asarray(mu, dtype=np.float64).reshape(-1)
    sigma = np.asarray(sigma, dtype=np.float64).reshape(-1)
    if acq == _lib.ACQ_SIGMA:
        return sigma
    if acq == _lib.ACQ_UCB:
        return sf * mu + param * sigma
    out = np.zeros_like(mu)
    live = np.nonzero(sigma != 0)[0]
    s = sigma[live]
    gain = sf * (mu[live] - incumbent) - param
    z = gain / s
    if acq == _lib.ACQ_PI:
        out[live] = ndtr(z)
    else:
        out[live] = gain * ndtr(z) + s * (np.exp(-0.5 * z * z) / np.sqrt(2.0 * np.pi))
    return out


class AcquisitionFunction:
    """factory interface, turbo/modules/acquisition_functions.py:12-77"""

    def get_type(self):
        raise NotImplementedError()

    def construct_function(self, trial_num, model, desired_extremum, *args):
        raise NotImplementedError()

    class FunctionInstance:
        def __init__(self, model, desired_extremum):
            assert _is_native(model) or hasattr(model, 'predict'), 'not a Surrogate.ModelInstance: {!r}'.format(type(model))
            self.model = model
            assert desired_extremum in ('min', 'max')
            self.desired_extremum = desired_extremum
            self.scale_factor = 1 if desired_extremum == 'max' else -1

        def get_name(self):
            raise NotImplementedError()

        def _native_args(self):
            """(acq enum, incumbent, param)"""
            raise NotImplementedError()

        def __call__(self, X):
            acq, incumbent, param = self._native_args()
            if not _is_native(self.model):
                mu, sigma = self.model.predict(X, return_std_dev=True)
                return _from_mu_sigma(acq, self.scale_factor, incumbent, param, mu, sigma)
            res = self.model._sweep(X, acq, self.scale_factor, incumbent, param, want_acq=True,
                                    want_sigma=False)
            return res['acq']

        last_sweep_ms = None      # device time of the last maximise* call (hipEvents around the sweep)

        @property
        def sweep_dtype(self):
            """arithmetic of the model's candidate sweep: 'f64', or 'f32' / 'f32h2' / 'f32x3' (a foreign model: 'f64').
            Models of up to 256 points are ALWAYS swept in f64, whatever the factory's dtype: the one-workgroup kernels
            for N <= 128 and the one-launch sweep for 128 < N <= 256 (csrc/small_kernels.hip) only exist in f64 -- an
            upgrade, never a loss, and the caller need not re-form the winner's value in f64."""
            factory = getattr(self.model, '_factory', None)
            dtype = getattr(factory, 'dtype', 'f64')
            if dtype == 'f64':
                return 'f64'
            # the LIBRARY says which kernels the last sweep took (tgp_last_timings slot 6): the size rule alone is not the
            # whole condition -- TGP_SMALL=0 / TGP_MID=0, a device without the one-launch sweep's LDS opt-in, or a model
            # another one displaced send a small model down the general sweep in the handle's arithmetic
            ctx = getattr(factory, '_native', None)
            if ctx is not None and getattr(factory, '_resident', None) is self.model and hasattr(ctx, 'last_timings') \
                    and not getattr(ctx, 'host', False):
                f64 = ctx.last_timings().get('sweep_f64', -1)
                if f64 >= 0:
                    return 'f64' if f64 else dtype
            n_obs = getattr(getattr(self.model, 'X', None), 'shape', (1 << 30,))[0]
            return 'f64' if n_obs <= 256 else dtype

        def maximise(self, X):
            """arg-max over the rows of X: (index, value); lowest index wins ties"""
            acq, incumbent, param = self._native_args()
            if not _is_native(self.model):
                vals = self(X)
                vals = np.where(np.isnan(vals), -np.inf, vals)      # NaN never wins, as on the GPU
                i = int(np.argmax(vals))                            # first maximum = lowest index
                return i, float(vals[i])
            res = self.model._sweep(X, acq, self.scale_factor, incumbent, param)
            self.last_sweep_ms = res.get('sweep_ms')
            return res['best_idx'], res['best_val']

        def maximise_topk(self, X, k):
            """the k best rows of X: (indices (k,), values (k,)), best first, lowest index on ties;
            the (M,) acquisition vector stays on the GPU (``tgp_sweep_topk``)"""
            _require_native(self.model, 'maximise_topk')
            acq, incumbent, param = self._native_args()
            ctx = self.model._ensure_resident()
            ctx.set_candidates(X)
            idx, vals = ctx.sweep_topk(min(int(k), 64), acq, self.scale_factor, incumbent, param)
            keep = idx >= 0
            return idx[keep], vals[keep]

        def refine(self, starting_points, bounds, max_iter=200):
            """the gradient stage on the GPU (``tgp_acq_refine``): every restart refined by a projected
            L-BFGS, together (N <= 128: each in its own workgroup of one launch); returns
            (x (R, D), values (R,), evaluations of the slowest restart)"""
            _require_native(self.model, 'refine')
            import warnings
            acq, incumbent, param = self._native_args()
            ctx = self.model._ensure_resident()
            low, high = zip(*bounds)
            x, v, status, its = ctx.acq_refine(starting_points, low, high, acq, self.scale_factor,
                                               incumbent, param, max_iter)
            for j in np.nonzero(status == 0)[0]:
                warnings.warn('restart {} of the on-device optimisation stopped at max_iter'.format(j))
            return x, v, its

        def lbfgsb(self, starting_points, bounds, max_iter=15000):
            """the gradient stage as the reference runs it -- SciPy's L-BFGS-B from every start
            (turbo/modules/auxiliary_optimisers.py:80-99) -- inside the library (``tgp_acq_lbfgsb``): the restarts in
            lock-step, one batched closed-form value + gradient evaluation per round; returns
            (x (R, D), values (R,), status (R,): 1 = SciPy's success, evaluations)"""
            _require_native(self.model, 'lbfgsb')
            acq, incumbent, param = self._native_args()
            ctx = self.model._ensure_resident()
            low, high = zip(*bounds)
            return ctx.acq_refine(starting_points, low, high, acq, self.scale_factor, incumbent, param, max_iter,
                                  lbfgsb=True)

        def winner_record(self, global_offset):
            """Attach a device-resident (D + 2,) float64 record to the model's GPU context: every
            later sweep packs [best value, global_offset + best index, candidate row] into it on
            the GPU (``tgp_set_winner_out``).  Returns the torch tensor that owns the memory -- the
            input of the sharded arg-max's all-gather over RCCL."""
            _require_native(self.model, 'winner_record')
            import torch
            ctx = self.model._ensure_resident()
            D = self.model.X.shape[1]
            rec = getattr(ctx, '_winner_keepalive', None)
            if rec is None or rec.numel() != D + 2:
                rec = torch.zeros(D + 2, dtype=torch.float64, device='cuda:%d' % ctx.device)
            ctx.set_winner_out(rec.data_ptr(), global_offset, keepalive=rec)
            return rec

        def value_and_grad(self, X):
            """acquisition values (m,) and their gradients (m, D) at a small batch of points,
            in closed form on the GPU (the reference differentiates 1-point calls by finite
            differences: turbo/modules/auxiliary_optimisers.py:80-92)"""
            _require_native(self.model, 'value_and_grad')
            acq, incumbent, param = self._native_args()
            ctx = self.model._ensure_resident()
            return ctx.acq_grad(X, acq, self.scale_factor, incumbent, param)

        def maximise_host_stream(self, num_points, low, high, topk=0, first=0, count=None):
            """The reference's HOST draw (``random_selector``: NumPy's global RNG, a column per parameter) made resident
            without forming the batch on the host (``tgp_set_candidates_mt19937``: the generator's sequential part in
            the library, the doubles on the GPU; ``np.random`` ends where NumPy's own calls would leave it), then swept.
            Returns None -- nothing drawn -- where the library cannot promise NumPy's numbers; otherwise
            ``(ctx, best_index, best_value, top)`` with ``top = (indices, values)`` of the ``topk`` best (None for 0);
            rows of the batch come from ``ctx.get_candidate``.  ``first`` / ``count``: only rows [first, first + count) of
            the ``num_points``-row batch are kept and swept (a rank's shard; indices are local to it) while ``np.random``
            ends behind the whole batch."""
            if not _is_native(self.model):
                return None
            acq, incumbent, param = self._native_args()
            ctx = self.model._ensure_resident()
            if not hasattr(ctx, 'set_candidates_numpy_stream') or not ctx.set_candidates_numpy_stream(num_points, low, high, first=first, count=count):
                return None
            if topk > 0:
                idx, vals = ctx.sweep_topk(min(int(topk), 64), acq, self.scale_factor, incumbent, param)
                keep = idx >= 0
                top = (idx[keep], vals[keep])
                if len(top[0]) > 0:
                    return ctx, int(top[0][0]), float(top[1][0]), top
                return ctx, 0, -np.inf, top      # nothing ranked (an all-NaN batch): index 0, as maximise reports it
            res = ctx.sweep(acq, self.scale_factor, incumbent, param)
            self.last_sweep_ms = res.get('sweep_ms')
            return ctx, res['best_idx'], res['best_val'], None

        def maximise_generated(self, num_points, low, high, seed, first_candidate=0, lhs_total=None, prefetch_seed=None):
            """draw `num_points` candidates in [low, high) on the GPU -- independent uniform ones, or
            (lhs_total given) rows first_candidate.. of an lhs_total-point Latin hypercube design --
            and return the best: (x (D,), value, index).  Candidates never cross PCIe.

            ``prefetch_seed``: after this sweep, draw the batch of the NEXT call (same shape, that seed) right away and
            arm ``tgp_set_overlap``: the candidates do not depend on the model, so the next trial's fit
            (turbo/optimiser.py:336) starts their sweep inside itself -- candidate scaling, cross-kernel and the first
            row tiles of the contraction beside the Cholesky's panel chain -- and the next call of this method finds
            the batch resident and skips the draw.  Same candidates, same values, same winner as without it."""
            _require_native(self.model, 'maximise_generated')
            acq, incumbent, param = self._native_args()
            ctx = self.model._ensure_resident()

            def draw(sd):
                if lhs_total is not None:
                    ctx.gen_candidates_lhs(sd, first_candidate, num_points, lhs_total, low, high)
                else:
                    ctx.gen_candidates(sd, first_candidate, num_points, low, high)

            lo_b, hi_b = np.asarray(low, dtype=np.float64).tobytes(), np.asarray(high, dtype=np.float64).tobytes()
            key = ("lhs" if lhs_total is not None else "uniform", int(seed), int(first_candidate), int(num_points),
                   int(lhs_total) if lhs_total is not None else None, lo_b, hi_b)
            if getattr(ctx, 'gen_key', None) != key:      # (resident already when the previous call prefetched it)
                draw(seed)
            res = ctx.sweep(acq, self.scale_factor, incumbent, param)
            self.last_sweep_ms = res.get('sweep_ms')
            best = ctx.get_candidate(res['best_idx']), res['best_val'], res['best_idx']
            if prefetch_seed is not None and hasattr(ctx, 'set_overlap') and not getattr(ctx, 'host', False):
                draw(prefetch_seed)
                ctx.prefetched = True      # ModelInstance._ensure_resident arms tgp_set_overlap for the next fit
            return best


class UCB(AcquisitionFunction):
    def __init__(self, beta):
        """beta: a constant float or a function of the trial number (:81-87)"""
        self.beta = beta

    def get_type(self):
        return 'optimism'

    def construct_function(self, trial_num, model, desired_extremum):
        beta = self.beta(trial_num) if callable(self.beta) else self.beta
        acq_info = {'beta': beta}
        return UCB.FunctionInstance(model, desired_extremum, beta), acq_info

    class FunctionInstance(AcquisitionFunction.FunctionInstance):
        """sf * mu + beta * sigma;  beta = inf -> sigma  (:147-158)"""

        def __init__(self, model, desired_extremum, beta):
            super().__init__(model, desired_extremum)
            self.beta = beta

        def get_name(self):
            return 'UCB' if self.desired_extremum == 'max' else '-LCB'

        def _native_args(self):
            if isinf(self.beta):
                return _lib.ACQ_SIGMA, 0.0, 0.0
            return _lib.ACQ_UCB, 0.0, self.beta


class PI(AcquisitionFunction):
    def __init__(self, xi):
        """xi: a constant float or a function of the trial number (:164-170)"""
        self.xi = xi

    def get_type(self):
        return 'improvement'

    def construct_function(self, trial_num, model, desired_extremum, incumbent_cost):
        xi = self.xi(trial_num) if callable(self.xi) else self.xi
        acq_info = {'xi': xi}
        return PI.FunctionInstance(model, desired_extremum, incumbent_cost, xi), acq_info

    class FunctionInstance(AcquisitionFunction.FunctionInstance):
        """Phi((sf * (mu - f+) - xi) / sigma), 0 where sigma == 0  (:225-247)"""

        def __init__(self, model, desired_extremum, incumbent_cost, xi):
            super().__init__(model, desired_extremum)
            self.incumbent_cost = incumbent_cost
            self.xi = xi

        def get_name(self):
            return 'PI'

        def _native_args(self):
            return _lib.ACQ_PI, self.incumbent_cost, self.xi


class EI(AcquisitionFunction):
    def __init__(self, xi):
        """xi: a constant float or a function of the trial number (:251-257)"""
        self.xi = xi

    def get_type(self):
        return 'improvement'

    def construct_function(self, trial_num, model, desired_extremum, incumbent_cost):
        xi = self.xi(trial_num) if callable(self.xi) else self.xi
        acq_info = {'xi': xi}
        return EI.FunctionInstance(model, desired_extremum, incumbent_cost, xi), acq_info

    class FunctionInstance(AcquisitionFunction.FunctionInstance):
        """diff * Phi(Z) + sigma * phi(Z), 0 where sigma == 0  (:336-358)"""

        def __init__(self, model, desired_extremum, incumbent_cost, xi):
            super().__init__(model, desired_extremum)
            self.incumbent_cost = incumbent_cost
            self.xi = xi

        def get_name(self):
            return 'EI'

        def _native_args(self):
            return _lib.ACQ_EI, self.incumbent_cost, self.xi
