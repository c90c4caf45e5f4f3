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
    assert (s0, s1) == ((1001, 2), (1000, 2))          # ONE batch of 2001 rows: ceil(2001 / 2), then the rest
    assert x0 == x1 and i0 == i1                       # identical winner everywhere
    assert i0["shards"] == 2 and np.isfinite(i0["max_acq"])
    assert abs(x0[0][0] - 0.25) < 0.1 and abs(x0[0][1] - 0.75) < 0.1
    assert t0 == t1 == (1.0, [[1.0, 0.0]], 9)          # tie -> lowest global index (rank 1's 9)
    assert n0 == n1 == (-5.0, 1)


def _bench_worker(rank, world, port, q):
    """bench.py's own step function (fit + sweep + winner exchange) on a scaled-down config, in
    both sharding modes; the GPU context is the oracle-backed stand-in (no GPU here)"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from oracle_context import OracleBackedContext
    cfg = dict(bench.CONFIGS["c1"], N=48, M=1001)      # odd M: the last shard is one row shorter
    X, y, ls = bench.synth_train(cfg)
    inc = float(y.min())
    out = {}
    for weak in (False, True):
        Xc, m_local, offset, m_job = bench.shard_candidates(cfg, rank, world, weak)
        gp = OracleBackedContext()
        gp.fit(X, y, cfg["kind"], 1.0, ls, cfg["noise"], 1e-10, True)
        gp.set_candidates(Xc)
        step = bench.build_step(gp, cfg, X, y, ls, inc, world, offset, None, "gloo")
        r = step()
        out["weak" if weak else "strong"] = dict(
            m_local=m_local, offset=offset, m_job=m_job, local=(r["best_idx"], r["best_val"]),
            job=(r["job_best_idx"], r["job_best_val"], r["job_best_row"].tolist()))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_step_strong_and_weak_world2():
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    from oracle import gp_oracle as o
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = dict(bench.CONFIGS["c1"], N=48, M=1001)
    X, y, ls = bench.synth_train(cfg)
    om = o.fit(X, y, cfg["kind"], 1.0, ls, cfg["noise"], 1e-10, True)
    # strong: ONE batch of M rows (the 1-GPU run's batch) in contiguous shards of ceil(M / 2)
    s0, s1 = res[0]["strong"], res[1]["strong"]
    assert (s0["m_local"], s0["offset"], s0["m_job"]) == (501, 0, 1001)
    assert (s1["m_local"], s1["offset"], s1["m_job"]) == (500, 501, 1001)
    assert s0["job"] == s1["job"]                                   # same winner on every rank
    whole = bench.synth(cfg, 0, cfg["M"])[2]
    acq, wi, wv = o.sweep(om, whole, cfg["acq"], "min", cfg["param"], float(y.min()))
    assert s0["job"][0] == wi and s0["job"][1] == wv                # == the single-GPU arg-max
    assert s0["job"][2] == [whole[wi].tolist()]
    # weak: M rows of its own per rank, global index = rank * M + local index
    w0, w1 = res[0]["weak"], res[1]["weak"]
    assert (w0["m_local"], w0["offset"], w0["m_job"]) == (1001, 0, 2002)
    assert (w1["m_local"], w1["offset"], w1["m_job"]) == (1001, 1001, 2002)
    assert w0["job"] == w1["job"]
    locals_ = [(w0["local"][1], w0["local"][0]), (w1["local"][1], 1001 + w1["local"][0])]
    best = max(locals_, key=lambda t: (t[0], -t[1]))
    assert (w0["job"][1], w0["job"][0]) == best
    assert w0["local"] == s0["local"] or s0["m_local"] != w0["m_local"]   # rank 0's weak stream is THE batch


def _bench8_worker(rank, world, port, q):
    """bench.py's step at world size 8: M not divisible by 8 (ragged last shard) and M < 8 (empty
    shards), strong and weak"""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    torch.set_num_threads(1)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from oracle_context import OracleBackedContext
    out = {}
    for M in (1003, 5):
        cfg = dict(bench.CONFIGS["c1"], N=40, M=M)
        X, y, ls = bench.synth_train(cfg)
        inc = float(y.min())
        for weak in (False, True):
            Xc, m_local, offset, m_job = bench.shard_candidates(cfg, rank, world, weak)
            gp = OracleBackedContext()
            gp.fit(X, y, cfg["kind"], 1.0, ls, cfg["noise"], 1e-10, True)
            if m_local > 0:
                gp.set_candidates(Xc)
            step = bench.build_step(gp, cfg, X, y, ls, inc, world, offset, None, "gloo", m_local)
            r = step()
            out[(M, weak)] = dict(m_local=m_local, offset=offset, m_job=m_job,
                                  job=(r["job_best_idx"], r["job_best_val"], np.asarray(r["job_best_row"]).tolist()))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_step_world8_ragged_and_empty_shards():
    """the 8-rank run rehearsed before the first real one: rank arithmetic, the
    exchange with ranks that hold nothing, the same winner everywhere = the single-GPU arg-max"""
    import torch.multiprocessing as mp
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import bench
    from oracle import gp_oracle as o
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench8_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for M in (1003, 5):
        cfg = dict(bench.CONFIGS["c1"], N=40, M=M)
        X, y, ls = bench.synth_train(cfg)
        om = o.fit(X, y, cfg["kind"], 1.0, ls, cfg["noise"], 1e-10, True)
        whole = bench.synth(cfg, 0, M)[2]
        acq, wi, wv = o.sweep(om, whole, cfg["acq"], "min", cfg["param"], float(y.min()))
        per = -(-M // 8)
        for r in range(8):
            s = res[r][(M, False)]
            lo = min(r * per, M)
            assert (s["m_local"], s["offset"], s["m_job"]) == (min(lo + per, M) - lo, lo, M)
            assert s["job"] == res[0][(M, False)]["job"]
            w = res[r][(M, True)]
            assert (w["m_local"], w["offset"], w["m_job"]) == (M, r * M, 8 * M)
            assert w["job"] == res[0][(M, True)]["job"]
        assert sum(res[r][(M, False)]["m_local"] for r in range(8)) == M
        if M == 5:
            assert [res[r][(M, False)]["m_local"] for r in range(8)] == [1, 1, 1, 1, 1, 0, 0, 0]
        job = res[0][(M, False)]["job"]
        # == the single-GPU arg-max (the oracle's BLAS sums a 1-row shard in another order than the batch: 1e-14)
        assert job[0] == wi and job[1] == pytest.approx(wv, rel=1e-12) and job[2] == [whole[wi].tolist()]
        assert 0 <= res[0][(M, True)]["job"][0] < 8 * M


def _mixed_worker(rank, world, port, q):
    """rank 1's shard is EMPTY (one candidate for two ranks) and the job is won by the gradient stage,
    so the exchange mixes the branches: host-held winners with indices past the batch"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import turbo_amd as ta

    class Acq:
        calls = 0

        def __call__(self, X):
            return -((X[:, 0] - 0.3) ** 2 + (X[:, 1] - 0.6) ** 2)

        def maximise(self, X):
            v = self(X)
            return int(np.argmax(v)), float(v.max())

        def value_and_grad(self, X):
            Acq.calls += 1
            return self(X), np.stack([-2 * (X[:, 0] - 0.3), -2 * (X[:, 1] - 0.6)], axis=1)
    b = ta.Bounds([("a", 0.0, 1.0), ("b", 0.0, 1.0)])
    np.random.seed(50 + rank)
    sizes = []

    def gen(n, lb):
        sizes.append(n)
        return np.random.uniform(0, 1, (n, 2))
    x, info = ta.CandidateSweep(num_random=1, grad_restarts=3, start_from_best=0, gen_random=gen)(b, Acq())
    q.put((rank, x.tolist(), info["max_acq"], info["best_global_index"], info["shards"], sizes))
    dist.barrier()
    dist.destroy_process_group()


def test_empty_shard_and_gradient_stage_winner_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mixed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, x0, v0, g0, s0, z0), (r1, x1, v1, g1, s1, z1) = res
    assert z0 == [1, 3] and z1 == [3]                  # rank 1 draws no candidates, only its restarts' starts
    assert x0 == x1 and v0 == v1 and g0 == g1 and s0 == s1 == 2
    assert v0 == pytest.approx(0.0, abs=1e-9) and np.allclose(x0, [[0.3, 0.6]], atol=1e-5)
    assert g0 >= 1                                     # a gradient-stage winner: index past the batch of 1


def test_shard_plan_rules():
    from turbo_amd.distributed import shard_plan
    assert [shard_plan(10, 4, r) for r in range(4)] == [(3, 0, 10), (3, 3, 10), (3, 6, 10), (1, 9, 10)]
    assert [shard_plan(3, 4, r) for r in range(4)] == [(1, 0, 3), (1, 1, 3), (1, 2, 3), (0, 3, 3)]
    assert shard_plan(262144, 8, 5) == (32768, 163840, 262144)
    assert shard_plan(100, 1, 0) == (100, 0, 100)
    assert shard_plan(100, 4, 2, weak=True) == (100, 200, 400)


def test_reduce_winners_rules():
    from turbo_amd.distributed import reduce_winners
    assert reduce_winners([0.5, 0.7, 0.7], [0, 20, 10]) == 2
    assert reduce_winners([float("nan"), -1.0], [0, 1]) == 1
    assert reduce_winners([float("nan"), float("nan")], [0, 1]) == 0
    assert reduce_winners([3.0], [7]) == 0


def _one_batch_worker(rank, world, port, q):
    """the DEFAULT host draw across ranks: the same script -- the same np.random.seed -- on every rank"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import turbo_amd as ta

    seen = []

    class Acq:                      # a foreign acquisition: the batch is formed on the host
        def maximise(self, X):
            seen.append(np.array(X))
            v = -((X[:, 0] - 0.3) ** 2 + (X[:, 1] + 1.0) ** 2 + (X[:, 2] - 5.0) ** 2)
            i = int(np.argmax(v))
            return i, float(v[i])
    b = ta.Bounds([("a", 0.0, 1.0), ("b", -2.0, 3.0), ("c", 4.0, 4.5)])
    out = {}
    for M in (2001, 40001, 1):      # (40 001 x 3 takes the library's continuation of the stream, 2001 x 3 NumPy's loop; 1: an empty shard)
        seen.clear()
        np.random.seed(77)          # the SAME seed everywhere
        x, info = ta.CandidateSweep(num_random=M)(b, Acq())
        shard = seen[0] if seen else np.empty((0, 3))
        out[M] = (x.tolist(), info["max_acq"], shard.tolist() if M < 5000 else [shard.shape, float(shard.sum())], np.random.uniform(size=3).tolist())
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_default_host_draw_is_one_batch_across_ranks_world2():
    """ranks seeded alike sweep DISJOINT shards of the batch one process would have drawn, choose the point one process
    would have chosen, and leave np.random where one process would have left it (round 6; before, each rank drew the
    same ceil(M / G) numbers and the job swept half the batch twice)"""
    import torch.multiprocessing as mp
    import turbo_amd as ta
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_one_batch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, o0), (_, o1) = res
    lo, hi = [0.0, -2.0, 4.0], [1.0, 3.0, 4.5]
    for M in (2001, 40001, 1):
        np.random.seed(77)
        whole = np.hstack([np.random.uniform(a, b, size=(M, 1)) for a, b in zip(lo, hi)])
        tail = np.random.uniform(size=3).tolist()
        v = -((whole[:, 0] - 0.3) ** 2 + (whole[:, 1] + 1.0) ** 2 + (whole[:, 2] - 5.0) ** 2)
        i = int(np.argmax(v))
        per = -(-M // 2)
        for rank, o in ((0, o0), (1, o1)):
            x, max_acq, shard, after = o[M]
            assert x == [whole[i].tolist()] and max_acq == float(v[i]), (M, rank)      # the single-process winner, on every rank
            assert after == tail, (M, rank)                                            # np.random behind the WHOLE batch
            want = whole[rank * per:min((rank + 1) * per, M)]
            if M < 5000:
                assert shard == want.tolist(), (M, rank)
            else:
                assert shard[0] == want.shape and shard[1] == float(want.sum()), (M, rank)
