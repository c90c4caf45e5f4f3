"""Auxiliary optimiser: maximise the acquisition over a batch of M uniform candidates.

Mirror of stage 1 of the reference's ``RandomAndQuasiNewton``
(turbo/modules/auxiliary_optimisers.py:16-129: random stage :59-66, result :114-129) and of
``random_selector`` (turbo/modules/naive_selectors.py:39-46).  Call contract:
``aux_optimiser(latent_bounds, acq) -> (x (1, D), {'max_acq': float})`` (turbo/optimiser.py:340).

With a native acquisition instance the whole sweep (cross-kernel, triangular contraction,
acquisition, arg-max) is one call into libturbogp.so and only the winning (value, index) comes
back.  When ``torch.distributed`` is initialised with more than one rank, every rank sweeps its
own shard of the batch and the winners are combined with one all-gather (RCCL on GPUs).

The gradient stage (auxiliary_optimisers.py:69-112) is mirrored too: L-BFGS-B (SciPy, as in the
reference) from the ``start_from_best`` best random candidates plus fresh random starts.  With a
native acquisition instance the gradient comes in closed form from the GPU (``tgp_acq_grad``)
instead of finite differences over 1-point calls, and the restarts advance in LOCK-STEP: each
L-BFGS-B run lives in its own thread and asks for f(x); once every still-running restart has
asked, one ``tgp_acq_grad`` call serves them all (k points per call instead of k calls).  Every
restart sees exactly the values it would see alone, so the result equals the sequential loop.
"""
import threading
import warnings

import numpy as np

from .distributed import allgather_argmax, dist_info


class random_selector:
    """points uniform-random in the latent space: one column per parameter from the global
    NumPy RNG, hstacked (turbo/modules/naive_selectors.py:39-46)"""

    def __call__(self, num_points, latent_bounds):
        cols = []
        for name, pmin, pmax in latent_bounds.ordered:
            cols.append(np.random.uniform(pmin, pmax, size=(num_points, 1)))
        return np.hstack(cols)


class _Lockstep:
    """Rendezvous of k optimiser threads around one batched evaluator ``fn(X (m, D)) -> (v, g)``."""

    def __init__(self, fn, n):
        self.fn = fn
        self.cv = threading.Condition()
        self.active = n
        self.pending = {}
        self.results = {}
        self.error = None
        self.batches = []          # points per batched call (observability / tests)

    def _flush(self):
        # called with the lock held, by whichever thread completed the round
        keys = sorted(self.pending)
        X = np.vstack([self.pending[k] for k in keys])
        self.pending.clear()
        try:
            v, g = self.fn(X)
            v, g = np.asarray(v, dtype=np.float64).reshape(-1), np.asarray(g, dtype=np.float64)
            for i, k in enumerate(keys):
                self.results[k] = (float(v[i]), g[i].copy())
            self.batches.append(len(keys))
        except BaseException as e:      # every waiting restart must wake up and fail
            self.error = e
            for k in keys:
                self.results[k] = None
        self.cv.notify_all()

    def request(self, j, x):
        with self.cv:
            if self.error is not None:
                raise self.error
            self.pending[j] = np.array(x, dtype=np.float64).reshape(1, -1)
            if len(self.pending) == self.active:
                self._flush()
            else:
                self.cv.wait_for(lambda: j in self.results)
            r = self.results.pop(j)
            if r is None:
                raise self.error
            return r

    def done(self, j):
        with self.cv:
            self.active -= 1
            if self.pending and len(self.pending) == self.active:
                self._flush()


class CandidateSweep:
    def __init__(self, num_random=1000, grad_restarts=0, start_from_best=0, gen_random=None,
                 shard=True, device_rng_seed=None, lockstep=True):
        """
        Args:
            num_random: number of random points to sample to search for the maximum
                (whole job; each rank takes ceil(num_random / world_size) when sharded)
            grad_restarts: number of restarts of the gradient-based optimiser (0 = sweep only)
            start_from_best: how many of those start from the best points of the random stage
                (should be <= num_random and <= grad_restarts)
            gen_random: candidate generator ``(num_points, latent_bounds) -> (M, D)``;
                defaults to ``random_selector()``
            shard: split the batch over the ranks of torch.distributed when initialised
            device_rng_seed: None (default) draws candidates like the reference, on the host
                from the global NumPy RNG.  An integer draws them ON the GPU instead (Philox
                stream seed + call number; shards are disjoint pieces of one stream), so the
                batch never crosses PCIe.  Needs a native acquisition instance.
            lockstep: run the gradient restarts in lock-step over batched gradient calls when the
                acquisition instance offers ``value_and_grad`` (False: one after the other)
        """
        assert num_random > 0, 'the candidate sweep needs num_random > 0'
        assert start_from_best <= num_random
        assert start_from_best <= grad_restarts
        self.num_random = num_random
        self.grad_restarts = grad_restarts
        self.start_from_best = start_from_best
        self.gen_random = gen_random or random_selector()
        self.shard = shard
        self.device_rng_seed = device_rng_seed
        self.lockstep = lockstep
        self.last_batches = None
        self._calls = 0

    def __call__(self, latent_bounds, acq):
        """Returns: x (1, num_attribs) within the bounds, {'max_acq': value}"""
        bounds = [(lb[1], lb[2]) for lb in latent_bounds.ordered]
        maximisation_info = {}
        rank, world = dist_info() if self.shard else (0, 1)
        m_local = -(-self.num_random // world)

        if self.device_rng_seed is not None:
            assert hasattr(acq, 'maximise_generated'), 'device_rng_seed needs a native acquisition'
            low, high = zip(*bounds)
            best_x, best_y, best_i = acq.maximise_generated(
                m_local, low, high, self.device_rng_seed + self._calls, first_candidate=rank * m_local)
            self._calls += 1
            best_x = np.asarray(best_x, dtype=np.float64).reshape(1, -1)
            random_x = None
        else:
            random_x = self.gen_random(m_local, latent_bounds)
        random_y = None
        if random_x is None:
            pass
        elif hasattr(acq, 'maximise') and not (self.grad_restarts > 0 and self.start_from_best > 0):
            best_i, best_y = acq.maximise(random_x)
        else:
            # a foreign acquisition callable: same argsort/[0] semantics as the reference
            # (auxiliary_optimisers.py:61-66), NaNs last
            random_y = -np.asarray(acq(random_x))
            best_i = int(np.argsort(random_y, axis=0, kind='stable').flatten()[0])
            best_y = float(-random_y[best_i])
        if random_x is not None:
            best_x = np.asarray(random_x[best_i], dtype=np.float64).reshape(1, -1)

        # minimise by gradient-based optimiser (auxiliary_optimisers.py:69-112)
        if self.grad_restarts > 0:
            all_warnings = []
            n_best = self.start_from_best if random_y is not None else 0
            starts = []
            if n_best > 0:
                order = np.argsort(random_y, axis=0, kind='stable').flatten()
                starts.append(random_x[order[:n_best]])
            if self.grad_restarts - n_best > 0:
                starts.append(self.gen_random(self.grad_restarts - n_best, latent_bounds))
            starting_points = np.vstack(starts)
            if self.lockstep and hasattr(acq, 'value_and_grad') and self.grad_restarts > 1:
                with warnings.catch_warnings(record=True) as ws:
                    warnings.simplefilter('always')
                    results = self._bfgs_lockstep(acq, starting_points, bounds)
                all_warnings.extend(ws)
            else:
                results = []
                for j in range(self.grad_restarts):
                    with warnings.catch_warnings(record=True) as ws:
                        warnings.simplefilter('always')
                        results.append(self._bfgs(acq, starting_points[j], bounds, j))
                    all_warnings.extend(ws)
            for j, (res_x, res_y) in enumerate(results):   # in restart order, as the reference's loop
                if res_y is not None and -res_y > best_y:
                    best_x = np.asarray(res_x, dtype=np.float64).reshape(1, -1)
                    best_y = -res_y
                    best_i = -1 - j     # not a member of the random batch
            if len(all_warnings) > 0:
                maximisation_info.update({'warnings': [w.message for w in all_warnings]})

        if world > 1:
            best_y, best_x, owner = allgather_argmax(best_y, best_x, rank * m_local + best_i)
            maximisation_info['shards'] = world
            maximisation_info['best_global_index'] = owner

        # ensure that the chosen value lies within the bounds (auxiliary_optimisers.py:120-124)
        low_bounds, high_bounds = zip(*bounds)
        best_x = np.clip(best_x, low_bounds, high_bounds)
        maximisation_info.update({'max_acq': float(best_y)})
        return best_x, maximisation_info


    def _minimise(self, neg_f, jac, starting_point, bounds, j):
        import scipy.optimize
        x0 = np.asarray(starting_point, dtype=np.float64).reshape(-1)
        result = scipy.optimize.minimize(fun=neg_f, x0=x0, jac=jac, bounds=bounds, method='L-BFGS-B',
                                         options=dict(maxiter=15000))
        if not result.success:
            warnings.warn('restart {}/{} of gradient-based optimisation failed'.format(
                j, self.grad_restarts))
            return None, None
        return result.x, float(result.fun)

    def _bfgs(self, acq, starting_point, bounds, j):
        """one L-BFGS-B run on -acq (auxiliary_optimisers.py:80-99); (x, fun) or (None, None)"""
        if hasattr(acq, 'value_and_grad'):
            def neg_f(x):
                v, g = acq.value_and_grad(x.reshape(1, -1))
                return -float(v[0]), -g[0]
            jac = True
        else:
            def neg_f(x):
                return -float(np.asarray(acq(x.reshape(1, -1))).reshape(-1)[0])
            jac = None
        return self._minimise(neg_f, jac, starting_point, bounds, j)

    def _bfgs_lockstep(self, acq, starting_points, bounds):
        """all restarts at once: one thread per L-BFGS-B run, one batched gradient call per round"""
        n = len(starting_points)
        sync = _Lockstep(acq.value_and_grad, n)
        out = [(None, None)] * n
        errors = [None] * n

        def run(j):
            def neg_f(x):
                v, g = sync.request(j, x)
                return -v, -g
            try:
                out[j] = self._minimise(neg_f, True, starting_points[j], bounds, j)
            except BaseException as e:
                errors[j] = e
            finally:
                sync.done(j)

        threads = [threading.Thread(target=run, args=(j,), name='tgp-restart-%d' % j) for j in range(n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        self.last_batches = sync.batches
        for e in errors:
            if e is not None:
                raise e
        return out


# the reference's name for this slot, so presets written against it keep working
RandomAndQuasiNewton = CandidateSweep
