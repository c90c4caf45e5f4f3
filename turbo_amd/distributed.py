"""Sharded arg-max over the ranks of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the
CPU tests).  The candidate batch shards with no data-path collective; the only exchange is the
winner of each shard: ``D + 2`` doubles per rank, one all-gather, then every rank reduces the
``world`` winners locally with the same rule (largest value, lowest global index on ties) so all
ranks return the same point.  RCCL has no MAXLOC reduction, hence gather-then-reduce.
"""
import numpy as np


def _dist():
    try:
        import torch.distributed as dist
    except Exception:   # torch absent: single process
        return None
    if dist.is_available() and dist.is_initialized():
        return dist
    return None


def dist_info():
    """(rank, world_size); (0, 1) when torch.distributed is not initialised"""
    d = _dist()
    if d is None:
        return 0, 1
    return d.get_rank(), d.get_world_size()


def dist_backend(group=None):
    """'nccl' (= RCCL), 'gloo', ... or None when torch.distributed is not initialised"""
    d = _dist()
    return None if d is None else str(d.get_backend(group))


def reduce_winners(vals, idxs):
    """index into the gathered lists of the winning shard: max value, then lowest global index;
    NaN never wins unless everything is NaN"""
    best = 0
    for r in range(1, len(vals)):
        v, b = vals[r], vals[best]
        better = (v > b) or (np.isnan(b) and not np.isnan(v)) or (v == b and idxs[r] < idxs[best])
        if better:
            best = r
    return best


def allgather_records(rec, group=None, ctx=None):
    """All-gather one packed winner record per rank and reduce.

    ``rec``: torch float64 tensor ``[value, global index, row (D)]`` that already lives where the
    backend wants it (the GPU for RCCL -- e.g. the buffer ``tgp_set_winner_out`` fills -- or the
    host for gloo).  One collective, one device-to-host copy of ``world * (D + 2)`` doubles.
    Returns (value, row (1, D), global_index), identical on every rank.

    Ordering: the record is WRITTEN by libturbogp.so on the library's own non-blocking stream and
    READ here by RCCL on torch's current stream.  ``ctx`` (the ``NativeGP`` whose sweep packed ``rec``)
    makes that explicit: ``tgp_winner_wait`` puts a wait for the event the library recorded behind the
    packing kernel onto torch's current stream before the collective is issued (round 6).  Without
    ``ctx`` -- a record built on the host, gloo -- there is no second stream to order."""
    import torch
    d = _dist()
    if ctx is not None and rec.is_cuda and hasattr(ctx, 'winner_wait'):
        ctx.winner_wait(torch.cuda.current_stream(rec.device).cuda_stream)
    world = d.get_world_size(group)
    out = torch.empty(world * rec.numel(), dtype=torch.float64, device=rec.device)
    d.all_gather_into_tensor(out, rec, group=group)
    allv = out.cpu().numpy().reshape(world, -1)
    w = reduce_winners(list(allv[:, 0]), list(allv[:, 1].astype(np.int64)))
    return float(allv[w, 0]), allv[w, 2:].reshape(1, -1).copy(), int(allv[w, 1])


def allgather_argmax(value, row, global_index, group=None):
    """Combine per-rank winners held on the HOST.  Returns (value, row (1, D), global_index) of
    the overall winner, identical on every rank."""
    import torch
    d = _dist()
    row = np.asarray(row, dtype=np.float64).reshape(-1)
    if d is None or d.get_world_size(group) == 1:
        return float(value), row.reshape(1, -1), int(global_index)
    backend = d.get_backend(group)
    dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    # the index travels as a float64: exact below 2**53
    mine = torch.tensor(np.concatenate([[float(value), float(global_index)], row]),
                        dtype=torch.float64, device=dev)
    return allgather_records(mine, group=group)


def shard_plan(m_total, world, rank, weak=False):
    """Contiguous candidate shards.  strong (default): a FIXED batch of m_total
    rows cut into ceil(m_total / world) rows per rank, the last ranks' shards shorter or empty;
    weak: every rank sweeps m_total rows of its own.  Returns (m_local, global_offset,
    m_job) with global_offset = global index of this rank's row 0."""
    if weak:
        return int(m_total), int(rank) * int(m_total), int(m_total) * int(world)
    per = -(-int(m_total) // int(world))
    lo = min(int(rank) * per, int(m_total))
    hi = min(lo + per, int(m_total))
    return hi - lo, lo, int(m_total)
