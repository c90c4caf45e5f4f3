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
instead of finite differences over 1-point calls.
"""
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


class CandidateSweep:
    def __init__(self, num_random=1000, grad_restarts=0, start_from_best=0, gen_random=None,
                 shard=True, device_rng_seed=None):
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
            for j in range(self.grad_restarts):
                with warnings.catch_warnings(record=True) as ws:
                    warnings.simplefilter('always')
                    res_x, res_y = self._bfgs(acq, starting_points[j], bounds, j)
                all_warnings.extend(ws)
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


    def _bfgs(self, acq, starting_point, bounds, j):
        """one L-BFGS-B run on -acq (auxiliary_optimisers.py:80-99); (x, fun) or (None, None)"""
        import scipy.optimize
        x0 = np.asarray(starting_point, dtype=np.float64).reshape(-1)
        if hasattr(acq, 'value_and_grad'):
            def neg_f(x):
                v, g = acq.value_and_grad(x.reshape(1, -1))
                return -float(v[0]), -g[0]
            jac = True
        else:
            def neg_f(x):
                return -float(np.asarray(acq(x.reshape(1, -1))).reshape(-1)[0])
            jac = None
        result = scipy.optimize.minimize(fun=neg_f, x0=x0, jac=jac, bounds=bounds, method='L-BFGS-B',
                                         options=dict(maxiter=15000))
        if not result.success:
            warnings.warn('restart {}/{} of gradient-based optimisation failed'.format(
                j, self.grad_restarts))
            return None, None
        return result.x, float(result.fun)


# the reference's name for this slot, so presets written against it keep working
RandomAndQuasiNewton = CandidateSweep
