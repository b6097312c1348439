"""Replica-ensemble sharding across the GPUs of one node (one process per GPU, torch.distributed).

The hot path shards over replicas only: replicas are independent, so there is NO collective on the
step path.  Each rank owns a contiguous block of global replica ids -- the id is what keys the RNG and selects the
demand scenario, so results do not depend on the number of ranks.  RCCL (backend "nccl" on ROCm; "gloo" on CPU for the
tests) is used only to combine results after stepping:

  * ``ensemble_moments``          all_reduce(sum) of per-link first/second moments -> ensemble mean / variance
  * ``gather_replica_summaries``  all_gather of small per-replica vectors (observations, rewards, totals)

Message sizes are a few KB to a few hundred KB (2 x L x 8 B moments; R x k x 4 B summaries): latency-bound, so a single
flat collective per call is used and nothing is bucketed.
"""
import numpy as np


def shard(total_replicas: int, world_size: int, rank: int):
    """Contiguous block of global replica ids owned by ``rank``: (offset, count).  Blocks differ by at most one."""
    if not 0 <= rank < world_size:
        raise ValueError("rank outside world")
    base, extra = divmod(int(total_replicas), int(world_size))
    count = base + (1 if rank < extra else 0)
    offset = rank * base + min(rank, extra)
    return offset, count


def _dist():
    import torch.distributed as dist

    return dist if dist.is_available() and dist.is_initialized() else None


def _to_tensor(a, device):
    import torch

    return torch.as_tensor(np.ascontiguousarray(a), device=device)


def _device():
    import torch

    dist = _dist()
    if dist is not None and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def ensemble_moments(local_values: np.ndarray):
    """local_values [R_local, ...] -> (count, mean, variance) over ALL ranks' replicas (population variance)."""
    v = np.asarray(local_values, dtype=np.float64)
    n = np.array([v.shape[0]], dtype=np.float64)
    s1, s2 = v.sum(axis=0), (v * v).sum(axis=0)
    dist = _dist()
    if dist is not None and dist.get_world_size() > 1:
        dev = _device()
        packed = _to_tensor(np.concatenate([n, s1.ravel(), s2.ravel()]), dev)
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
        packed = packed.cpu().numpy()
        k = s1.size
        n, s1, s2 = packed[:1], packed[1:1 + k].reshape(s1.shape), packed[1 + k:].reshape(s2.shape)
    mean = s1 / n[0]
    var = np.maximum(s2 / n[0] - mean * mean, 0.0)
    return int(n[0]), mean, var


def gather_replica_summaries(local: np.ndarray, total_replicas: int = None):
    """local [R_local, k] -> [R_total, k] on every rank, rows ordered by global replica id (ranks own contiguous blocks,
    possibly of different sizes)."""
    a = np.ascontiguousarray(local)
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return a
    import torch

    world, dev = dist.get_world_size(), _device()
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    counts[dist.get_rank()] = a.shape[0]
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    counts = counts.cpu().tolist()
    mx = max(counts)
    pad = np.zeros((mx,) + a.shape[1:], dtype=a.dtype)
    pad[:a.shape[0]] = a
    mine = _to_tensor(pad, dev)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = np.concatenate([p.cpu().numpy()[:c] for p, c in zip(parts, counts)], axis=0)
    if total_replicas is not None and out.shape[0] != total_replicas:
        raise RuntimeError(f"gathered {out.shape[0]} replicas, expected {total_replicas}")
    return out
