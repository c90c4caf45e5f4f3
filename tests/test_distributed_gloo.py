"""The N > 1 path on CPU: world_size 2 over gloo.  Each rank sweeps its own candidate shard
(stand-in acquisition, the GPU is not involved), winners are combined with one all-gather and
every rank must return the same point."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import turbo_amd as ta
    from turbo_amd.distributed import allgather_argmax, dist_info
    assert dist_info() == (rank, world)

    class Acq:
        def maximise(self, X):
            v = -((X[:, 0] - 0.25) ** 2 + (X[:, 1] - 0.75) ** 2)
            i = int(np.argmax(v))
            return i, float(v[i])
    b = ta.Bounds([("a", 0.0, 1.0), ("b", 0.0, 1.0)])
    np.random.seed(100 + rank)          # every rank draws its own shard
    shard = None

    def gen(n, lb):
        nonlocal shard
        shard = ta.random_selector()(n, lb)
        return shard
    x, info = ta.CandidateSweep(num_random=2001, gen_random=gen)(b, Acq())
    # tie across ranks -> lowest global index wins, on every rank
    v, row, gi = allgather_argmax(1.0, np.array([float(rank), 0.0]), 10 - rank)
    # NaN never beats a number
    v2, _, gi2 = allgather_argmax(float("nan") if rank == 0 else -5.0, np.zeros(2), rank)
    q.put((rank, x.tolist(), info, shard.shape, (v, row.tolist(), gi), (v2, gi2)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_argmax_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, x0, i0, s0, t0, n0), (r1, x1, i1, s1, t1, n1) = res
    assert s0 == s1 == (1001, 2)                       # ceil(2001 / 2) candidates per rank
    assert x0 == x1 and i0 == i1                       # identical winner everywhere
    assert i0["shards"] == 2 and np.isfinite(i0["max_acq"])
    assert abs(x0[0][0] - 0.25) < 0.1 and abs(x0[0][1] - 0.75) < 0.1
    assert t0 == t1 == (1.0, [[1.0, 0.0]], 9)          # tie -> lowest global index (rank 1's 9)
    assert n0 == n1 == (-5.0, 1)


def test_reduce_winners_rules():
    from turbo_amd.distributed import reduce_winners
    assert reduce_winners([0.5, 0.7, 0.7], [0, 20, 10]) == 2
    assert reduce_winners([float("nan"), -1.0], [0, 1]) == 1
    assert reduce_winners([float("nan"), float("nan")], [0, 1]) == 0
    assert reduce_winners([3.0], [7]) == 0
