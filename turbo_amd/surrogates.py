"""Surrogate plugin: a GP fitted and queried on one MI355X through libturbogp.so.

Mirrors the reference's plugin contract so an unmodified ``turbo.Optimiser`` can use it as
``optimiser.surrogate`` (turbo/optimiser.py:43-47, :336):

    Surrogate / Surrogate.ModelInstance          turbo/modules/surrogates.py:22-81
    SciKitGPSurrogate (the semantics mirrored)   turbo/modules/surrogates.py:225-365

Hyper-parameters are either fixed (``model_params['optimizer'] = None`` or
``training_iterations = 0``) or fitted like scikit-learn does it for the reference
(sklearn/gaussian_process/_gpr.py:296-337): L-BFGS-B on the negative log marginal likelihood in
log space, started from the kernel's values (or the previous trial's, ``param_continuity``) and
from ``iterations - 1`` random restarts.  Every objective evaluation -- kernel matrix, Cholesky,
K^-1 and the gradient trace -- runs on the GPU (``tgp_fit_grad``); SciPy only drives theta.
"""
import copy
import numbers
import warnings

import numpy as np

from . import _lib
from .kernels import GPKernel



_START_THREADS = None


def _start_threads():
    """the threads that drive the concurrent starts of a hyper-parameter fit (one per worker handle of the device's
    pool, at most four), created once per process"""
    global _START_THREADS
    if _START_THREADS is None:
        from concurrent.futures import ThreadPoolExecutor
        _START_THREADS = ThreadPoolExecutor(max_workers=4, thread_name_prefix='turbo_amd-start')
    return _START_THREADS


class Surrogate:
    """A probabilistic model factory: one fitted ModelInstance per trial
    (turbo/modules/surrogates.py:22-81)."""

    def construct_model(self, trial_num, X, y):
        """Returns: (model, fitting_info)"""
        raise NotImplementedError()

    class ModelInstance:
        def predict(self, X, return_std_dev=False):
            raise NotImplementedError()

        def get_hyper_params(self):
            raise NotImplementedError()

        def get_hyper_param_names(self):
            raise NotImplementedError()

        def get_log_likelihood(self):
            raise NotImplementedError()


class HipGPSurrogate(Surrogate):
    """Drop-in for ``SciKitGPSurrogate`` with the arithmetic on the GPU.

    ``model_params`` takes the same keys the reference forwards to
    ``GaussianProcessRegressor`` (turbo/modules/surrogates.py:245-251, :315):
    ``kernel`` (a ``GPKernel`` or a scikit-learn kernel object), ``alpha`` (jitter, default
    1e-10), ``normalize_y`` (default True as in the reference's defaults, :231-243),
    ``optimizer`` (None = fixed hyper-parameters; 'fmin_l_bfgs_b' (default): L-BFGS-B as scikit-learn runs it,
    walked INSIDE the library (``tgp_fit_lbfgsb``: SciPy's algorithm restated in C++, a thread and a stream per
    start driving the GPU objective with no interpreter between two evaluations -- the same iterates, evaluation
    counts and optimum as SciPy's on the same objective; unbounded theta falls back to 'scipy'); 'scipy': SciPy's own L-BFGS-B drives the GPU objective from Python (the default of
    rounds 1-4; ``parallel_restarts_above`` applies); 'device' (opt-in): ``tgp_fit_optimise`` -- N <= 128: every
    start side by side in ONE launch with a projected L-BFGS whose iterates are not SciPy's, its optima the same
    or better; larger problems: as the default; or a callable with scikit-learn's optimizer signature),
    ``random_state`` and ``n_restarts_optimizer``.
    """

    default_model_params = {
        # 1.0 * Matern(nu=2.5) + WhiteKernel()  (turbo/modules/surrogates.py:236)
        'kernel': GPKernel('matern52', 1.0, 1.0, 1.0),
        'normalize_y': True,
    }

    def __init__(self, model_params=None, training_iterations=None, param_continuity=True,
                 dtype='f64', device=0, incremental=True, parallel_restarts_above='auto'):
        """
        Args:
            model_params (dict): see class docstring
            training_iterations (None, int, or function of trial_num): as in the reference
                (surrogates.py:245-292).  With fixed hyper-parameters the value only ends up
                in ``fitting_info['iterations']``.
            param_continuity (bool): kept for signature compatibility (surrogates.py:269-271)
            dtype: 'f64', 'f32', or 'f32h2' / 'f32x3' (f32 accuracy from two scaled fp16 / three bf16
                planes on the 16-bit matrix pipe, opt-in) -- arithmetic of the candidate sweep; the fit (and the
                hyper-parameter optimisation) is always f64
            device: HIP device index
            incremental: when consecutive trials keep the hyper-parameters and only append one
                observation (the Optimiser's loop, turbo/optimiser.py:335-336), extend the
                resident factorisation by one row in O(N^2) instead of refitting in O(N^3)
            parallel_restarts_above: when the starts of the hyper-parameter fit (the warm start and the
                ``iterations - 1`` restarts) run side by side, one host thread and one GPU handle on a private
                stream each -- same iterates, same result as one after the other, bit for bit.  'auto'
                (default): where that was measured to pay, 64 < N <= 8192 on three threads (to 1536 until late in
                round 5; beyond, the workers' workspaces -- 4 N^2 doubles each -- set the limit): one evaluation
                there is a serial chain that leaves the chip idle and SciPy's own per-evaluation overhead of
                one start hides behind another start's kernels (round 4, three starts: N = 500 10.9 -> 6.5 ms,
                700 14.7 -> 7.8, 1000 27.8 -> 22.8, 1500 61 -> 57; N = 2048: no gain.  Round 5, with the start
                threads kept alive between fits -- starting three threads cost 1.2 ms per fit: N = 200 6.2 -> 4.2-5.5 ms,
                256 6.3 -> 4.9-6.5, 500 7.1 -> 5.1-6.0, 1000 23.4 -> 20.4-22.0 (host-bound: +-15 % run to run);
                N = 100 / 128 6.2 / 5.2 -> 5.8 / 4.8; below that the interpreter is the bottleneck and threads lose:
                N = 32 3.7 -> 5.8).  None: never.  A
                number: with more observations than that.
        """
        _lib.load()   # fail loudly, now, when the native library is missing ...
        if _lib.HOST_ONLY:
            # ... or when only the host-only build could be loaded: that one serves RELOADED models
            # (unpickling does not come through here); a new factory is the optimisation path and needs the GPU
            raise _lib.TurboGPLibraryError(
                "the GPU build of libturbogp.so could not be loaded ({}); this process loaded the host-only "
                "library, which can evaluate reloaded models but not fit new ones".format(_lib.LOAD_ERROR))
        self.model_params = model_params or self.default_model_params
        self.training_iterations = training_iterations
        assert training_iterations is None or self.model_params.get('n_restarts_optimizer') is None, \
            'cannot specify n_restarts_optimizer and training_iterations at the same time'
        self.param_continuity = param_continuity
        self.dtype = dtype
        self.device = device
        self.incremental = incremental
        self.parallel_restarts_above = parallel_restarts_above
        self.restart_threads = 3
        self._workers = []       # GPU contexts of the hyper-parameter fit's concurrent starts, while that fit runs
        self.last_worker_count = 0
        self._native = None      # one GPU context shared by every model this factory makes
        self._resident = None    # id of the model whose fit currently lives in the context
        self._last_model_params = None
        self._reloaded = False   # True once this factory came out of a pickle (Recorder.load_compressed)

    def _context(self):
        if self._native is None:
            try:
                self._native = _lib.NativeGP(self.device, self.dtype)
            except _lib.NoDeviceError:     # (also what the host-only library answers for any GPU device)
                # Only a RELOADED factory may go on without a GPU: the recorder's models are often
                # queried by the plot path in another process (turbo/recorder.py:157-163,
                # turbo/plotting/trials.py:192-195, :574-577) that owns no MI355X.  libturbogp.so's host
                # backend (csrc/host_backend.cpp) then answers predict / acquisition in float64.
                # A factory made in this process still fails loudly: the optimisation loop is GPU-only.
                if not getattr(self, '_reloaded', False):
                    raise
                warnings.warn('no HIP device in this process: the reloaded model is evaluated by the host '
                              'backend of libturbogp.so (float64; meant for plots, not for optimisation runs)')
                self._native = _lib.NativeGP(_lib.DEVICE_HOST, 'f64')
        return self._native

    def _get_training_iterations(self, trial_num):
        # turbo/modules/surrogates.py:283-292
        if self.training_iterations is None:
            iterations = self.model_params.get('n_restarts_optimizer')  # may be absent / None
        elif callable(self.training_iterations):
            iterations = self.training_iterations(trial_num)
        else:
            iterations = self.training_iterations
        assert iterations is not None, 'must specify the number of training iterations'
        assert iterations >= 0, 'invalid number of iterations: {}'.format(iterations)
        return iterations

    def construct_model(self, trial_num, X, y):
        """turbo/modules/surrogates.py:294-326"""
        iterations = self._get_training_iterations(trial_num)
        fitting_info = {'iterations': iterations}
        assert 'kernel' in self.model_params, 'you must specify a kernel for the GP'
        # don't want the initial parameter values to be changed, so make a copy
        kernel = GPKernel.from_any(self.model_params['kernel']).copy()
        optimizer = self.model_params.get('optimizer', 'fmin_l_bfgs_b')
        jitter = self.model_params.get('alpha', 1e-10)
        assert np.isscalar(jitter), 'only a scalar alpha is supported'
        normalize_y = bool(self.model_params.get('normalize_y', False))

        if self.param_continuity and self._last_model_params is not None:
            # theta is log-transformed (surrogates.py:302-304)
            kernel.theta = np.log(self._last_model_params.copy())

        # inputs are owned by the caller and may be mutated after return: copy on entry
        # (sklearn copy_X_train=True, _gpr.py:293-294)
        X = np.array(X, dtype=np.float64, copy=True, order='C')
        y = np.array(y, dtype=np.float64, copy=True).reshape(-1)
        assert X.ndim == 2 and X.shape[0] == y.shape[0], 'X must be (N, D) and y (N,)'
        if kernel.anisotropic:
            assert len(kernel.length_scale) == X.shape[1], \
                'anisotropic length scale needs one entry per dimension'

        with warnings.catch_warnings(record=True) as ws:
            warnings.simplefilter('always')
            trained = iterations > 0 and optimizer is not None and len(kernel.theta) > 0
            if trained:
                evals = self._optimise(kernel, X, y, float(jitter), normalize_y, optimizer,
                                       iterations - 1)   # for scikit: 0 restarts => 1 iteration
                fitting_info.update({'lml_evaluations': evals})
            elif iterations == 0:
                fitting_info.update({'fixed': kernel.theta})
            model = HipGPSurrogate.ModelInstance(self, X, y, kernel, float(jitter), normalize_y)
            model._ensure_resident()
        if len(ws) > 0:
            fitting_info.update({'warnings': [w.message for w in ws]})
        if trained:
            # theta is log-transformed (surrogates.py:322-324)
            self._last_model_params = np.exp(kernel.theta.copy())
        fitting_info.update({'fit_ms': model.fit_ms})
        return model, fitting_info

    def _rng(self):
        # sklearn.utils.check_random_state(self.random_state), created per fit (_gpr.py:253)
        seed = self.model_params.get('random_state', None)
        if seed is None or seed is np.random:
            return np.random.mtrand._rand
        if isinstance(seed, numbers.Integral):
            return np.random.RandomState(seed)
        return seed

    def _optimise(self, kernel, X, y, jitter, normalize_y, optimizer, n_restarts):
        """maximise the log marginal likelihood over theta = log(hyper-parameters); mirrors
        GaussianProcessRegressor.fit (_gpr.py:296-337) and _constrained_optimization (:654-670).
        Leaves the best theta in ``kernel``; returns the number of GPU objective evaluations."""
        import scipy.optimize
        ctx = self._context()
        self._resident = None        # the context is about to hold other hyper-parameters
        count = [0]

        def obj_func(theta, eval_gradient=True):
            kernel.theta = theta
            count[0] += 1
            try:
                lml, grad = ctx.fit_grad(X, y, kernel.kind, kernel.constant, kernel.length_scale,
                                         kernel.noise_level, jitter, normalize_y)
            except np.linalg.LinAlgError:
                # _gpr.py:586-589: not PD -> -inf likelihood, zero gradient
                return (np.inf, np.zeros_like(theta)) if eval_gradient else np.inf
            g = kernel.select_gradient(grad)
            return (-lml, -g) if eval_gradient else -lml

        def constrained_optimization(initial_theta, bounds):
            if optimizer == 'fmin_l_bfgs_b':
                res = scipy.optimize.minimize(obj_func, initial_theta, method='L-BFGS-B', jac=True,
                                              bounds=bounds)
                if res.status != 0:
                    warnings.warn('lbfgs failed to converge (status={}): {}'.format(res.status, res.message))
                return res.x, res.fun
            elif callable(optimizer):
                return optimizer(obj_func, initial_theta, bounds=bounds)
            raise ValueError('Unknown optimizer {}.'.format(optimizer))

        bounds = kernel.theta_bounds
        if optimizer in ('device', 'fmin_l_bfgs_b'):
            done = self._optimise_in_library(ctx, kernel, X, y, jitter, normalize_y, bounds, n_restarts,
                                             lbfgsb=optimizer == 'fmin_l_bfgs_b')
            if done is not None:
                return done
        if optimizer in ('device', 'scipy'):
            optimizer = 'fmin_l_bfgs_b'      # (or: unbounded theta) SciPy drives tgp_fit_grad
        starts = [kernel.theta.copy()]
        if n_restarts > 0:
            if not np.isfinite(bounds).all():
                raise ValueError('Multiple optimizer restarts (n_restarts_optimizer>0) requires '
                                 'that all bounds are finite.')
            rng = self._rng()
            for _ in range(n_restarts):
                starts.append(rng.uniform(bounds[:, 0], bounds[:, 1]))
        n_obs = X.shape[0]
        if self.parallel_restarts_above == 'auto':
            side_by_side, threads = 64 < n_obs <= 8192, 3
        elif self.parallel_restarts_above is None:
            side_by_side, threads = False, 1
        else:
            side_by_side, threads = n_obs > self.parallel_restarts_above, self.restart_threads
        if len(starts) > 1 and side_by_side and optimizer == 'fmin_l_bfgs_b':
            # Above the small-problem sizes one evaluation leaves most of the chip idle (the fit is a
            # serial panel chain): the starts run side by side, one host thread and one handle on a
            # private stream each.  Every start walks exactly the iterates it walks alone (its own
            # L-BFGS-B, its own handle), and scikit-learn draws the restarts' initial points
            # independently of the earlier results (_gpr.py:326-330), so the outcome is unchanged.
            optima = self._optimise_starts_in_threads(kernel, X, y, jitter, normalize_y, bounds, starts, count, threads)
        else:
            optima = [constrained_optimization(t0, bounds) for t0 in starts]
        best = int(np.argmin([o[1] for o in optima]))
        kernel.theta = optima[best][0]
        return count[0]

    def _optimise_starts_in_threads(self, kernel, X, y, jitter, normalize_y, bounds, starts, count, threads):
        # One handle on a private stream per start, borrowed from the DEVICE's pool of worker handles for the duration
        # of this fit (tgp_workers_acquire; round 5).  Round 4 gave every factory three of its own and kept them for its
        # lifetime (creating a stream costs the runtime ~1 ms): with two factories alive in a process the runtime's
        # hardware queues were over-subscribed -- two streams on one queue run one after the other -- and a
        # hyper-parameter fit took twice as long (N = 500: 11.6 -> 19.3 ms).  The pool keeps the streams alive too, but
        # there are at most four of them per device whatever the number of factories.  More starts than workers: the
        # starts take the workers in turn (start j on worker j mod n), each thread owning one worker.
        n = min(len(starts), threads, 4)
        with self._context().workers(n) as workers:
            self._workers = workers          # (what tools look at DURING the fit; the views are dead once the pool is released)
            self.last_worker_count = n       # ... and what stays behind: how many starts ran side by side
            try:
                return self._run_starts(workers, kernel, X, y, jitter, normalize_y, bounds, starts, count, n)
            finally:
                self._workers = []

    def _run_starts(self, workers, kernel, X, y, jitter, normalize_y, bounds, starts, count, threads):
        import scipy.optimize

        evals = [0] * len(starts)     # one cell per start: no shared read-modify-write between the threads

        def run_one(j, w):
            k = kernel.copy()

            def obj_func(theta):
                k.theta = theta
                evals[j] += 1
                try:
                    lml, grad = w.fit_grad(X, y, k.kind, k.constant, k.length_scale, k.noise_level, jitter, normalize_y)
                except np.linalg.LinAlgError:
                    return np.inf, np.zeros_like(theta)      # _gpr.py:586-589
                return -lml, -k.select_gradient(grad)

            res = scipy.optimize.minimize(obj_func, starts[j], method='L-BFGS-B', jac=True, bounds=bounds)
            return res.x, res.fun, res.status, res.message

        def run_share(t):                     # thread t owns worker t and walks the starts t, t + n, ... one after the other
            return [(j, run_one(j, workers[t])) for j in range(t, len(starts), len(workers))]

        # (the module's ONE pool of start threads: a ThreadPoolExecutor made per fit spent ~1.2 ms of a 6.6 ms fit at
        # N = 500 starting its three threads -- Thread.start waits for each to come up)
        shares = list(_start_threads().map(run_share, range(len(workers))))
        results = [r for _, r in sorted((jr for share in shares for jr in share), key=lambda jr: jr[0])]
        count[0] += sum(evals)
        for x, f, status, message in results:
            if status != 0:
                warnings.warn('lbfgs failed to converge (status={}): {}'.format(status, message))
        return [(x, f) for x, f, _, _ in results]

    def _optimise_in_library(self, ctx, kernel, X, y, jitter, normalize_y, bounds, n_restarts, lbfgsb):
        """every start (the current theta + n_restarts drawn as scikit-learn draws them, _gpr.py:326-330) optimised
        inside the library, no SciPy and no interpreter between two evaluations.  ``lbfgsb`` (the default optimizer):
        ``tgp_fit_lbfgsb``, L-BFGS-B itself -- SciPy's iterates; otherwise (optimizer='device') ``tgp_fit_optimise``:
        N <= 128 (D <= 64) in ONE launch, a workgroup per start and a projected L-BFGS, above that the same L-BFGS-B.
        The library's vector is log(constant, length scale(s), noise) in full; a FIXED hyper-parameter travels as an entry
        with lo == hi, which the library leaves out of the optimiser's vector as scikit-learn leaves it out of theta.
        A kernel without a noise term travels with that entry fixed at log 0.
        Returns the number of objective evaluations, or None where it does not apply (unbounded theta)."""
        n_ls = len(kernel.length_scale) if kernel.anisotropic else 1
        P = 2 + n_ls
        if (kernel.noise is not None and not kernel.noise > 0) or not np.isfinite(bounds).all():
            return None
        free = np.asarray(kernel.select_gradient(np.arange(P, dtype=np.float64)), dtype=np.int64)   # positions of theta's entries in the full vector
        if len(free) != len(kernel.theta) or len(free) == 0:
            return None
        with np.errstate(divide='ignore'):      # (no noise term: the entry is fixed at log 0)
            full = np.log(np.concatenate([[kernel.constant], np.atleast_1d(np.asarray(kernel.length_scale, dtype=np.float64)),
                                          [kernel.noise_level]]))
        full_bounds = np.stack([full, full], axis=1)
        full_bounds[free] = bounds
        starts = [kernel.theta.copy()]
        if n_restarts > 0:
            rng = self._rng()
            for _ in range(n_restarts):
                starts.append(rng.uniform(bounds[:, 0], bounds[:, 1]))
        full_starts = np.tile(full, (len(starts), 1))
        full_starts[:, free] = np.array(starts)
        theta, f, status, evals = ctx.fit_optimise(X, y, kernel.kind, full_starts, n_ls, full_bounds, jitter,
                                                   normalize_y, max_iter=15000 if lbfgsb else 500, lbfgsb=lbfgsb)
        if lbfgsb:
            for st in status:     # what scikit-learn's _check_optimize_result says of SciPy's status 1 / 2
                if st != 1:
                    warnings.warn('lbfgs failed to converge (status={}): {}'.format(
                        1 if st == 0 else 2, 'STOP: TOTAL NO. OF ITERATIONS REACHED LIMIT' if st == 0
                        else 'ABNORMAL_TERMINATION_IN_LNSRCH'))
        elif np.any(status != 1):
            warnings.warn('on-device L-BFGS did not converge for start(s) {} (status {})'.format(
                np.nonzero(status != 1)[0].tolist(), status[status != 1].tolist()))
        f = np.where(np.isfinite(f), f, np.inf)
        kernel.theta = theta[int(np.argmin(f))][free]
        return int(evals)

    def predict_many(self, models, X, return_std_dev=False):
        """``[m.predict(X, return_std_dev) for m in models]`` as ONE library call per size class for the
        models that are small enough (N <= 128, and 128 < N <= 256): the plot path walks the recorder's trials and predicts
        the same grid with every trial's model (turbo/plotting/trials.py:371,448,574-577;
        turbo/plotting/surrogates.py:23-24,61-65), which otherwise re-fits every stored model in
        turn.  Returns mus (T, M) [, sigmas (T, M)]; rows equal the per-model results (bit for bit for
        N <= 128; to rounding above: the batched fit factors 64 x 64 blocks in another order)."""
        X = np.asarray(X, dtype=np.float64)
        if X.ndim == 1:
            X = X.reshape(1, -1)
        T, M = len(models), X.shape[0]
        mus = np.empty((T, M))
        sig = np.empty((T, M)) if return_std_dev else None
        groups = {}
        for t, m in enumerate(models):
            native = isinstance(m, HipGPSurrogate.ModelInstance) and m.X.shape[1] == X.shape[1]
            # two size classes, each one library call: N <= 128 (small-problem kernels) and 128 < N <= 256.
            # (a FOREIGN model -- the reference's ModelInstance only carries .model -- has no .X: looked at only when native)
            size_class = None
            if native:
                n_obs = m.X.shape[0]
                size_class = 0 if n_obs <= 128 else (1 if n_obs <= 256 else None)
            key = (m.kernel.kind, bool(m.normalize_y), size_class) if size_class is not None else None
            groups.setdefault(key, []).append(t)
        ctx = self._context()
        for key, members in groups.items():
            if key is None or M == 0 or getattr(ctx, 'host', False):   # (the host backend has no batched entry)
                for t in members:                  # large or foreign models: one by one
                    r = models[t].predict(X, return_std_dev)
                    mus[t] = r[0] if return_std_dev else r
                    if return_std_dev:
                        sig[t] = r[1]
                continue
            specs = [dict(X=models[t].X, y=models[t].y, kind=key[0], constant=models[t].kernel.constant,
                          length_scale=models[t].kernel.length_scale, noise=models[t].kernel.noise_level,
                          jitter=models[t].jitter, normalize_y=key[1]) for t in members]
            mu, sg, lml, clamped = ctx.predict_batch(specs, X, want_sigma=return_std_dev)
            for k, t in enumerate(members):
                mus[t] = mu[k]
                if return_std_dev:
                    sig[t] = sg[k]
                if models[t].log_likelihood is None:
                    models[t].log_likelihood = float(lml[k])
            if clamped > 0 and return_std_dev:
                warnings.warn('Predicted variances smaller than 0. Setting those variances to 0.')
        return (mus, sig) if return_std_dev else mus

    def close(self):
        """release the GPU contexts this factory holds (its own and the concurrent starts' workers, each
        with N^2 fit buffers and a private stream); models re-create the main one on demand"""
        for w in self._workers:
            w.close()
        self._workers = []
        if self._native is not None:
            self._native.close()
            self._native = None
        self._resident = None

    # the GPU context is not picklable; models re-create it lazily (Recorder pickles models,
    # turbo/recorder.py:117-155)
    def __getstate__(self):
        d = dict(self.__dict__)
        d['_native'] = None
        d['_workers'] = []
        d['_resident'] = None
        return d

    def __setstate__(self, d):
        # ANY unpickle marks the factory as reloaded (Recorder.load_compressed, but also copy.deepcopy and a
        # multiprocessing hand-over): in a process without a HIP device such a factory answers predict /
        # acquisition calls from the host backend, with a warning, instead of raising.  In a process WITH a
        # GPU the flag changes nothing.
        self.__dict__.update(d)
        self._reloaded = True

    class ModelInstance(Surrogate.ModelInstance):
        """A GP fitted to one trial's data set.  Holds only host-side state (X, y, theta);
        the factorisation lives in the factory's GPU context and is rebuilt on demand when
        another model has displaced it (models are retained per trial by the Recorder:
        turbo/optimiser.py:343)."""

        def __init__(self, factory, X, y, kernel, jitter, normalize_y):
            self._factory = factory
            self.X = X
            self.y = y
            self.kernel = kernel
            self.jitter = jitter
            self.normalize_y = normalize_y
            self.log_likelihood = None
            self.y_mean = None
            self.y_std = None
            self.fit_ms = None
            self.appended = False

        # ---- native plumbing ----
        def _ensure_resident(self):
            f = self._factory
            ctx = f._context()
            if f._resident is not self:
                k = self.kernel
                ls = k.length_scale
                if np.ndim(ls) > 0:
                    assert len(ls) == self.X.shape[1], \
                        'anisotropic length scale needs one entry per dimension'
                if hasattr(ctx, 'set_overlap') and not getattr(ctx, 'host', False):
                    # the next sweep's batch is resident already (CandidateSweep(prefetch_next=True) drew it on the GPU
                    # behind the previous sweep): this fit starts that sweep inside itself.  Anything that replaced the
                    # resident batch since then cleared gen_key.
                    ctx.set_overlap(2 if (getattr(ctx, 'prefetched', False) and getattr(ctx, 'gen_key', None) is not None) else 0)
                lml, ym, ys = ctx.fit(self.X, self.y, k.kind, k.constant, ls, k.noise_level,
                                      self.jitter, self.normalize_y, append=f.incremental)
                self.appended = ctx.appended
                self.log_likelihood, self.y_mean, self.y_std = lml, ym, ys
                self.fit_ms = ctx.profile_read()['last_fit_ms']
                f._resident = self
            return ctx

        def _sweep(self, X, acq, sf=1.0, incumbent=0.0, param=0.0, want_mu=False,
                   want_sigma=False, want_acq=False):
            ctx = self._ensure_resident()
            X = np.asarray(X, dtype=np.float64)
            if X.ndim == 1 and X.size > 0:
                X = X.reshape(1, -1)
            assert X.ndim == 2 and X.shape[1] == self.X.shape[1], \
                'X must have shape (num_points, {})'.format(self.X.shape[1])
            if X.shape[0] == 0:      # an empty batch: empty results, nothing to launch
                e = np.empty(0)
                return dict(mu=e if want_mu else None, sigma=e.copy() if want_sigma else None,
                            acq=e.copy() if want_acq else None, best_val=float('nan'), best_idx=-1,
                            n_clamped=0)
            if hasattr(ctx, 'evaluate'):     # one call into the library (tgp_evaluate)
                res = ctx.evaluate(X, acq, sf, incumbent, param, want_mu, want_sigma, want_acq)
            else:
                ctx.set_candidates(X)
                res = ctx.sweep(acq, sf, incumbent, param, want_mu, want_sigma, want_acq)
            if res['n_clamped'] > 0 and want_sigma:
                # sklearn/gaussian_process/_gpr.py:480-485
                warnings.warn('Predicted variances smaller than 0. Setting those variances to 0.')
            return res

        # ---- Surrogate.ModelInstance ----
        def predict(self, X, return_std_dev=False):
            """turbo/modules/surrogates.py:332-338: mus (M,) [, sigmas (M,)]"""
            res = self._sweep(X, _lib.ACQ_NONE, want_mu=True, want_sigma=return_std_dev)
            if return_std_dev:
                return res['mu'], res['sigma']
            return res['mu']

        def get_hyper_params(self):
            return self.kernel.hyper_params()

        def get_hyper_param_names(self):
            return self.kernel.hyper_param_names()

        def get_log_likelihood(self):
            if self.log_likelihood is None:
                self._ensure_resident()
            return self.log_likelihood

        def __getstate__(self):
            return dict(self.__dict__)
