"""Auxiliary optimiser: maximise the acquisition over a batch of M uniform candidates.

Mirror of stage 1 of the reference's ``RandomAndQuasiNewton``
(turbo/modules/auxiliary_optimisers.py:16-129: random stage :59-66, result :114-129) and of
``random_selector`` (turbo/modules/naive_selectors.py:39-46).  Call contract:
``aux_optimiser(latent_bounds, acq) -> (x (1, D), {'max_acq': float})`` (turbo/optimiser.py:340).

With a native acquisition instance the whole sweep (cross-kernel, triangular contraction,
acquisition, arg-max) is one call into libturbogp.so and only the winning (value, index) comes
back.  When ``torch.distributed`` is initialised with more than one rank, every rank sweeps its
own shard of the batch and the winners are combined with one all-gather (RCCL on GPUs).

The gradient stage (auxiliary_optimisers.py:69-112) is a "next" row:
``grad_restarts`` other than 0 raises NotImplementedError.
"""
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
            grad_restarts, start_from_best: must be 0 (gradient stage not built yet)
            gen_random: candidate generator ``(num_points, latent_bounds) -> (M, D)``;
                defaults to ``random_selector()``
            shard: split the batch over the ranks of torch.distributed when initialised
            device_rng_seed: None (default) draws candidates like the reference, on the host
                from the global NumPy RNG.  An integer draws them ON the GPU instead (Philox
                stream seed + call number; shards are disjoint pieces of one stream), so the
                batch never crosses PCIe.  Needs a native acquisition instance.
        """
        assert num_random > 0, 'the candidate sweep needs num_random > 0'
        if grad_restarts != 0 or start_from_best != 0:
            raise NotImplementedError('the gradient-based stage is not built yet: use '
                                      'grad_restarts=0, start_from_best=0')
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
        if random_x is None:
            pass
        elif hasattr(acq, 'maximise'):
            best_i, best_y = acq.maximise(random_x)
        else:
            # a foreign acquisition callable: same argsort/[0] semantics as the reference
            # (auxiliary_optimisers.py:61-66), NaNs last
            random_y = -np.asarray(acq(random_x))
            best_i = int(np.argsort(random_y, axis=0, kind='stable').flatten()[0])
            best_y = float(-random_y[best_i])
        if random_x is not None:
            best_x = np.asarray(random_x[best_i], dtype=np.float64).reshape(1, -1)

        if world > 1:
            best_y, best_x, owner = allgather_argmax(best_y, best_x, rank * m_local + best_i)
            maximisation_info['shards'] = world
            maximisation_info['best_global_index'] = owner

        # ensure that the chosen value lies within the bounds (auxiliary_optimisers.py:120-124)
        low_bounds, high_bounds = zip(*bounds)
        best_x = np.clip(best_x, low_bounds, high_bounds)
        maximisation_info.update({'max_acq': float(best_y)})
        return best_x, maximisation_info


# the reference's name for this slot, so presets written against it keep working
RandomAndQuasiNewton = CandidateSweep
