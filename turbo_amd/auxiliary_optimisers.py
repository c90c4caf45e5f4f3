"""Auxiliary optimiser: maximise the acquisition over a batch of M uniform candidates.

Mirror of stage 1 of the reference's ``RandomAndQuasiNewton``
(turbo/modules/auxiliary_optimisers.py:16-129: random stage :59-66, result :114-129) and of
``random_selector`` (turbo/modules/naive_selectors.py:39-46).  Call contract:
``aux_optimiser(latent_bounds, acq) -> (x (1, D), {'max_acq': float})`` (turbo/optimiser.py:340).

With a native acquisition instance the whole sweep (cross-kernel, triangular contraction,
acquisition, arg-max) is one call into libturbogp.so and only the winning (value, index) comes
back.  When ``torch.distributed`` is initialised with more than one rank, every rank sweeps its
own shard of the batch and the winners are combined with one all-gather (RCCL on GPUs).  With the
default ``random_selector`` the shards are rows of the ONE batch NumPy's global RNG holds for the
whole job (ranks seeded alike sweep what a single process would sweep); a generator of the
caller's own is asked for the rank's row count and decides itself what those rows are.

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

from .naive_selectors import random_selector
from .distributed import allgather_argmax, allgather_records, dist_backend, dist_info, shard_plan


def _native_acq(acq):
    """False for one of OUR acquisition instances over a FOREIGN model (a Surrogate.ModelInstance that is not
    HipGPSurrogate's): its value_and_grad / refine / lbfgsb need the GPU model, so the gradient stage differentiates
    1-point calls by finite differences instead, as the reference does (auxiliary_optimisers.py:80-92)"""
    model = getattr(acq, 'model', None)
    return model is None or hasattr(model, '_sweep')     # (no .model at all: a duck-typed instance that brings its own methods)


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
    _warned_host_draw = False     # the large-host-draw hint is given once per process
    STREAM_DRAW_MIN = 1 << 17     # elements of a batch from which the default host draw is finished on the GPU

    def __init__(self, num_random=1000, grad_restarts=0, start_from_best=0, gen_random=None,
                 shard=True, device_rng_seed=None, lockstep=True, on_device=False, max_iter=200,
                 device_design='uniform', prefetch_next=False):
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
            lockstep: run the gradient restarts in lock-step over batched gradient calls (native acquisition
                instances).  True (default): L-BFGS-B walked inside the library (``tgp_acq_lbfgsb``: SciPy's algorithm in
                C++, one batched evaluation per round, each restart's walk the one SciPy would take); 'scipy': a
                Python thread and a SciPy L-BFGS-B per restart meeting at a rendezvous (rounds 2-4's default);
                False: SciPy, one restart after the other
            on_device: run the gradient stage as a batched projected L-BFGS ON the GPU
                (``tgp_acq_refine``: every restart resident; N <= 128 the whole stage in one launch,
                above one launch sequence per iteration for all restarts)
                instead of SciPy's L-BFGS-B on the host.  Needs a native acquisition instance.
            max_iter: iteration cap of the on-device optimiser
            prefetch_next: with ``device_rng_seed``: draw the NEXT call's batch right behind this call's sweep, so that
                the next trial's fit starts that batch's sweep inside itself (``tgp_set_overlap``: the candidates do
                not depend on the model; turbo/optimiser.py:336-340 runs fit and maximisation back to back).  The
                candidates, the values and the chosen points are those of ``prefetch_next=False``, bit for bit.
            device_design: with ``device_rng_seed``: 'uniform' (independent uniform candidates, the
                counterpart of ``random_selector``) or 'lhs' (the whole batch of ``num_random``
                candidates is one Latin hypercube design, the counterpart of ``LHS_selector``;
                shards are rows of that one design)
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
        self.on_device = on_device
        self.max_iter = max_iter
        assert device_design in ('uniform', 'lhs')
        self.device_design = device_design
        self.prefetch_next = bool(prefetch_next)
        self.last_batches = None
        self._calls = 0

    def __call__(self, latent_bounds, acq):
        """Returns: x (1, num_attribs) within the bounds, {'max_acq': value}"""
        bounds = [(lb[1], lb[2]) for lb in latent_bounds.ordered]
        maximisation_info = {}
        rank, world = dist_info() if self.shard else (0, 1)
        # contiguous shards of ONE batch of num_random candidates; global index =
        # offset + local index, so the tie rule (lowest index) does not depend on the world size
        m_local, offset, _ = shard_plan(self.num_random, world, rank)
        # with RCCL the winner record [value, global index, row] is packed on the GPU by the sweep
        # itself and all-gathered from there
        rec = None
        if world > 1 and m_local > 0 and hasattr(acq, 'winner_record') and _native_acq(acq) and dist_backend() == 'nccl':
            rec = acq.winner_record(offset)

        random_x = random_y = None
        best_x, best_y, best_i = None, -np.inf, 0
        # The DEFAULT host draw (random_selector: NumPy's global RNG) across ranks: every rank draws the WHOLE batch of
        # num_random rows and keeps rows [offset, offset + m_local) of it, so that ranks started from the same script --
        # the same np.random.seed -- sweep disjoint shards of the ONE batch a single GPU would have swept (and end with
        # the same RNG state), instead of each drawing the same m_local numbers.  (With seeds of their own the ranks
        # still get valid, different shards.)  A generator of the caller's own keeps its contract: it is asked for
        # m_local rows and decides itself what they are.
        one_batch = world > 1 and type(self.gen_random) is random_selector and self.device_rng_seed is None

        def draw_shard():
            if one_batch:
                return self.gen_random(self.num_random, latent_bounds)[offset:offset + m_local]
            return self.gen_random(m_local, latent_bounds)
        if m_local == 0:
            # more ranks than candidates: this rank only takes part in the exchange (and, on the default draw, passes
            # over the batch so that its RNG stays in step with the other ranks')
            if one_batch:
                self.gen_random(self.num_random, latent_bounds)
        elif self.device_rng_seed is not None:
            assert hasattr(acq, 'maximise_generated'), 'device_rng_seed needs a native acquisition'
            low, high = zip(*bounds)
            best_x, best_y, best_i = acq.maximise_generated(
                m_local, low, high, self.device_rng_seed + self._calls, first_candidate=offset,
                lhs_total=self.num_random if self.device_design == 'lhs' else None,
                prefetch_seed=(self.device_rng_seed + self._calls + 1) if self.prefetch_next else None)
            best_x = np.asarray(best_x, dtype=np.float64).reshape(1, -1)
        else:
            # The reference's draw (random_selector: NumPy's global RNG) for a native acquisition and a large batch:
            # only the generator's sequential recurrence runs on the host, inside the library; the GPU forms the
            # doubles and keeps the batch (tgp_set_candidates_mt19937) -- the numbers, the winner and np.random's
            # state afterwards are those of the host loop below, bit for bit (tests/test_gpu_host_stream.py)
            stream = None
            if (type(self.gen_random) is random_selector and hasattr(acq, 'maximise_host_stream') and _native_acq(acq)
                    and m_local * len(bounds) >= self.STREAM_DRAW_MIN
                    and not (self.grad_restarts > 0 and self.start_from_best > 64)):
                k = self.start_from_best if self.grad_restarts > 0 else 0
                low, high = zip(*bounds)
                try:
                    low, high = [float(v) for v in low], [float(v) for v in high]
                    if one_batch:
                        stream = acq.maximise_host_stream(self.num_random, low, high, topk=k, first=offset, count=m_local)
                    else:
                        stream = acq.maximise_host_stream(m_local, low, high, topk=k)
                except (TypeError, ValueError):
                    stream = None
            if stream is not None:
                ctx, best_i, best_y, top = stream
                rows_of = lambda order: np.vstack([ctx.get_candidate(int(i)) for i in order])   # noqa: E731
                if top is not None:
                    random_y = top
            elif m_local * len(bounds) > 1000000 and not CandidateSweep._warned_host_draw and hasattr(acq, 'maximise_generated') and _native_acq(acq):
                # once per process: a candidate generator of the caller's own at BASELINE's headline sizes -- the batch is
                # formed on the host and uploaded (NumPy's own loop for C3's 262 144 x 32: 45-66 ms for a 35 ms GPU step)
                CandidateSweep._warned_host_draw = True
                warnings.warn('CandidateSweep draws {} x {} candidates on the host with {}: at this size the draw and its upload cost '
                              'more than the GPU sweep -- the default random_selector (same numbers as the reference, finished on the '
                              'GPU) or device_rng_seed=<int> avoid that'.format(m_local, len(bounds), type(self.gen_random).__name__))
            if stream is not None:
                pass
            elif hasattr(acq, 'maximise') and not (self.grad_restarts > 0 and self.start_from_best > 0):
                random_x = draw_shard()
                best_i, best_y = acq.maximise(random_x)
            elif hasattr(acq, 'maximise_topk') and _native_acq(acq) and self.grad_restarts > 0 and self.start_from_best <= 64:
                # the best start_from_best candidates come back from the GPU (tgp_sweep_topk, k <= 64);
                # the (M,) acquisition vector stays there.  More starts than that take the branch below
                # (the vector comes back and is argsorted here, as the reference does).
                random_x = draw_shard()
                top_i, top_y = acq.maximise_topk(random_x, self.start_from_best)
                if len(top_i) > 0:
                    best_i, best_y = int(top_i[0]), float(top_y[0])
                else:   # nothing ranked (an all-NaN batch): index 0 as acq.maximise reports it
                    best_i, best_y = 0, -np.inf
                random_y = (top_i, top_y)
            else:
                # a foreign acquisition callable: same argsort/[0] semantics as the reference
                # (auxiliary_optimisers.py:61-66), NaNs last
                random_x = draw_shard()
                random_y = -np.asarray(acq(random_x))
                best_i = int(np.argsort(random_y, axis=0, kind='stable').flatten()[0])
                best_y = float(-random_y[best_i])
            if stream is None:
                rows_of = lambda order: random_x[order]   # noqa: E731
            best_x = np.asarray(rows_of([best_i]), dtype=np.float64).reshape(1, -1)
        self._calls += 1
        if hasattr(acq, 'last_sweep_ms') and acq.last_sweep_ms is not None:
            maximisation_info['sweep_ms'] = acq.last_sweep_ms
        from_sweep = True

        # minimise by gradient-based optimiser (auxiliary_optimisers.py:69-112)
        if self.grad_restarts > 0:
            all_warnings = []
            n_best = self.start_from_best if random_y is not None else 0
            starts = []
            if n_best > 0:
                if isinstance(random_y, tuple):
                    order = np.asarray(random_y[0], dtype=np.int64)[:n_best]
                else:
                    order = np.argsort(random_y, axis=0, kind='stable').flatten()[:n_best]
                n_best = len(order)     # (fewer when the batch ranked fewer: the rest start at random)
                starts.append(rows_of(order))
            if self.grad_restarts - n_best > 0:
                starts.append(self.gen_random(self.grad_restarts - n_best, latent_bounds))
            starting_points = np.vstack(starts)
            assert len(starting_points) == self.grad_restarts
            if self.on_device and hasattr(acq, 'refine') and _native_acq(acq):
                # all restarts advance together ON the GPU (tgp_acq_refine): no Python threads, no
                # SciPy, one kernel sequence per iteration for every restart
                with warnings.catch_warnings(record=True) as ws:
                    warnings.simplefilter('always')
                    xs, vs, its = acq.refine(starting_points, bounds, max_iter=self.max_iter)
                all_warnings.extend(ws)
                results = [(xs[j], -float(vs[j])) for j in range(len(vs))]
                maximisation_info['refine_iterations'] = int(its)
            elif self.lockstep and self.lockstep != 'scipy' and hasattr(acq, 'lbfgsb') and _native_acq(acq):
                # L-BFGS-B from every start as the reference runs it, walked inside the library (tgp_acq_lbfgsb): the
                # restarts in lock-step on one thread, one batched gradient evaluation per round, no interpreter
                # between two rounds -- each restart's walk is the one SciPy would take on the same objective
                with warnings.catch_warnings(record=True) as ws:
                    warnings.simplefilter('always')
                    xs, vs, status, evals = acq.lbfgsb(starting_points, bounds, max_iter=15000)
                    results = []
                    for j in range(len(vs)):
                        if status[j] != 1:          # SciPy's `not result.success` (auxiliary_optimisers.py:93-97)
                            warnings.warn('restart {}/{} of gradient-based optimisation failed'.format(j, self.grad_restarts))
                            results.append((None, None))
                        else:
                            results.append((xs[j], -float(vs[j])))
                all_warnings.extend(ws)
                maximisation_info['gradient_evaluations'] = int(evals)
            elif self.lockstep and hasattr(acq, 'value_and_grad') and _native_acq(acq) and self.grad_restarts > 1:
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
                    # not a member of the random batch: global indices past the batch, one block
                    # of grad_restarts per rank, so no two winners of a job share an index
                    best_i = (self.num_random - offset) + rank * self.grad_restarts + j
                    from_sweep = False
            if len(all_warnings) > 0:
                maximisation_info.update({'warnings': [w.message for w in all_warnings]})

        if world > 1:
            if rec is not None and from_sweep:
                best_y, best_x, owner = allgather_records(rec, ctx=getattr(getattr(acq.model, '_factory', None), '_native', None))
            else:
                if best_x is None:
                    best_x = np.zeros((1, len(bounds)))
                best_y, best_x, owner = allgather_argmax(best_y, best_x, offset + best_i)
            maximisation_info['shards'] = world
            maximisation_info['best_global_index'] = owner

        # A sweep in f32 arithmetic (dtype 'f32' / 'f32h2' / 'f32x3': BASELINE configs 3 and 4) ranks the batch
        # with a mean that carries the f32 rounding of the cross-kernel (|d mu| up to 1e-4 y_std at C3).  The
        # value REPORTED for the chosen point is formed once more in float64 -- cross-kernel row, mean, variance
        # and acquisition at that one point (tgp_acq_grad: closed form on the f64 factor) -- so max_acq meets
        # the fp64 bar; the sweep's own figure stays in the info.  Every rank holds the same model and, after
        # the exchange, the same point: the refined value is the same everywhere.
        if from_sweep and best_x is not None and getattr(acq, 'sweep_dtype', 'f64') != 'f64' and hasattr(acq, 'value_and_grad') and _native_acq(acq):
            v64, _ = acq.value_and_grad(np.asarray(best_x, dtype=np.float64).reshape(1, -1))
            if np.isfinite(v64[0]):
                maximisation_info['max_acq_sweep'] = float(best_y)
                best_y = float(v64[0])

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
        if hasattr(acq, 'value_and_grad') and _native_acq(acq):
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


class RandomAndQuasiNewton(CandidateSweep):
    """the reference's name for this slot WITH the reference's defaults (turbo/modules/auxiliary_optimisers.py:17:
    ``num_random=1000, grad_restarts=10, start_from_best=2``), so that code and presets written against it keep their
    meaning: ``RandomAndQuasiNewton()`` runs the gradient stage, ``CandidateSweep()`` is the pure sweep.  Everything
    else is ``CandidateSweep``."""

    def __init__(self, num_random=1000, grad_restarts=10, start_from_best=2, **kwargs):
        super().__init__(num_random=num_random, grad_restarts=grad_restarts, start_from_best=start_from_best, **kwargs)
