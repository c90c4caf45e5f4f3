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


def allgather_argmax(value, row, global_index, group=None):
    """Combine per-rank winners.  Returns (value, row (1, D), global_index) of the overall
    winner, identical on every rank."""
    import torch
    d = _dist()
    row = np.asarray(row, dtype=np.float64).reshape(-1)
    if d is None or d.get_world_size(group) == 1:
        return float(value), row.reshape(1, -1), int(global_index)
    world = d.get_world_size(group)
    backend = d.get_backend(group)
    dev = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' else torch.device('cpu')
    # the index travels as a float64: exact below 2**53
    mine = torch.tensor(np.concatenate([[float(value), float(global_index)], row]),
                        dtype=torch.float64, device=dev)
    out = torch.empty(world * mine.numel(), dtype=torch.float64, device=dev)
    d.all_gather_into_tensor(out, mine, group=group)
    allv = out.cpu().numpy().reshape(world, -1)
    w = reduce_winners(list(allv[:, 0]), list(allv[:, 1].astype(np.int64)))
    return float(allv[w, 0]), allv[w, 2:].reshape(1, -1).copy(), int(allv[w, 1])
