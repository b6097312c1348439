"""Replica-ensemble sharding across the GPUs of one node (one process per GPU, torch.distributed).

The hot path shards over replicas only: replicas are independent, so there is NO collective on the
step path.  Each rank owns a contiguous block of global replica ids -- the id is what keys the RNG and selects the
demand scenario, so results do not depend on the number of ranks.  RCCL (backend "nccl" on ROCm; "gloo" on CPU for the
tests) is used only to combine results after stepping:

  * ``ensemble_moments``          all_reduce(sum) of per-link first/second moments -> ensemble mean / variance
  * ``gather_replica_summaries``  all_gather of small per-replica vectors (observations, rewards, totals)

Message sizes are a few KB to a few hundred KB (2 x L x 8 B moments; R x k x 4 B summaries): latency-bound, so a single
flat collective per call is used and nothing is bucketed.

Both functions take numpy arrays (host results) or torch tensors.  A tensor stays on its device from end to end: with
the engine's observation / reward buffers (``VecPedNetEnv.step_device`` returns torch views of ``pedn_rl_device_ptr``) and
backend "nccl" the gather of config #5 (2048 x 20 float32 = 160 KB per step) is one RCCL all_gather over xGMI with no host
copy; ``VecPedNetEnv.gather_device`` wraps exactly that.
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


def _is_tensor(x):
    import sys

    torch = sys.modules.get("torch")
    return torch is not None and isinstance(x, torch.Tensor)


def _moments_tensor(v):
    """Device path of ``ensemble_moments``: sums, the all_reduce and the result stay on ``v``'s device."""
    import torch

    v = v.to(torch.float64)
    k = v[0].numel()
    packed = torch.cat([torch.tensor([float(v.shape[0])], dtype=torch.float64, device=v.device), v.sum(dim=0).reshape(-1),
                        (v * v).sum(dim=0).reshape(-1)])
    dist = _dist()
    if dist is not None and dist.get_world_size() > 1:
        dist.all_reduce(packed, op=dist.ReduceOp.SUM)
    n = packed[0]
    mean = (packed[1:1 + k] / n).reshape(v.shape[1:])
    var = torch.clamp(packed[1 + k:].reshape(v.shape[1:]) / n - mean * mean, min=0.0)
    return int(n.item()), mean, var


def _gather_tensor(a, total_replicas=None):
    """Device path of ``gather_replica_summaries``: one all_gather_into_tensor when every rank holds the same number of rows
    (the benchmark's and the RL configuration's case), a padded all_gather otherwise."""
    import torch

    a = a.contiguous()
    dist = _dist()
    if dist is None or dist.get_world_size() == 1:
        return a
    world = dist.get_world_size()
    counts = torch.zeros(world, dtype=torch.int64, device=a.device)
    counts[dist.get_rank()] = a.shape[0]
    dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    counts = counts.tolist()
    if min(counts) == max(counts):
        out = torch.empty((world * a.shape[0],) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
        dist.all_gather_into_tensor(out, a)
    else:
        mx = max(counts)
        pad = torch.zeros((mx,) + tuple(a.shape[1:]), dtype=a.dtype, device=a.device)
        pad[:a.shape[0]] = a
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        out = torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
    if total_replicas is not None and out.shape[0] != total_replicas:
        raise RuntimeError(f"gathered {out.shape[0]} replicas, expected {total_replicas}")
    return out


def ensemble_moments(local_values):
    """local_values [R_local, ...] -> (count, mean, variance) over ALL ranks' replicas (population variance).  numpy in ->
    numpy out; torch tensor in -> tensors on the same device out (no host copy)."""
    if _is_tensor(local_values):
        return _moments_tensor(local_values)
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


def gather_replica_summaries(local, total_replicas: int = None):
    """local [R_local, k] -> [R_total, k] on every rank, rows ordered by global replica id (ranks own contiguous blocks,
    possibly of different sizes).  numpy in -> numpy out; torch tensor in -> a tensor on the same device out (no host copy)."""
    if _is_tensor(local):
        return _gather_tensor(local, total_replicas)
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
