"""N > 1 path on CPU: two gloo ranks shard a replica ensemble, step their shards (with the CPU oracle standing in for
the engine -- there is no GPU here) and combine results with the collectives of pednstream_amd.ensemble; the outcome must
equal a single-process run over all replicas."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _densities(replica_ids, steps=40):
    import oracle_driver as od
    from golden_util import Golden, build_network
    from pednstream_amd.flatten import flatten_network

    net = build_network(Golden("nine_full"))     # speed noise 0.05: replicas differ from the first step
    model = flatten_network(net)
    rows = []
    for r in replica_ids:
        o = od.Oracle(model, seed=3, replica=r)
        o.run(1, steps)
        rows.append(o.field("speed")[:, steps - 1].astype(np.float64))
        o.close()
    return np.array(rows).reshape(len(replica_ids), model["n_links"])


def _worker(rank, world, port, total, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist

    from pednstream_amd import ensemble

    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    off, cnt = ensemble.shard(total, world, rank)
    local = _densities(range(off, off + cnt))
    n, mean, var = ensemble.ensemble_moments(local)
    gathered = ensemble.gather_replica_summaries(local, total_replicas=total)
    # the tensor path (what runs on the GPUs with device tensors): same collectives, nothing leaves the tensor's device
    import torch

    tn, tmean, tvar = ensemble.ensemble_moments(torch.as_tensor(local))
    tg = ensemble.gather_replica_summaries(torch.as_tensor(local), total_replicas=total)
    assert isinstance(tg, torch.Tensor) and tn == n
    assert np.array_equal(tg.numpy(), gathered) and np.allclose(tmean.numpy(), mean, rtol=1e-13) and np.allclose(tvar.numpy(), var, rtol=1e-9, atol=1e-15)
    even = torch.as_tensor(local[:2])          # equal row counts on both ranks: the all_gather_into_tensor branch
    ge = ensemble.gather_replica_summaries(even)
    assert ge.shape[0] == 2 * world and np.array_equal(ge[2 * rank:2 * rank + 2].numpy(), local[:2])
    q.put((rank, off, cnt, n, mean, var, gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_blocks_are_disjoint_and_complete():
    from pednstream_amd.ensemble import shard

    for total, world in ((4096, 8), (5, 2), (7, 8), (1024, 1)):
        blocks = [shard(total, world, r) for r in range(world)]
        ids = [i for off, cnt in blocks for i in range(off, off + cnt)]
        assert ids == list(range(total))
        assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    with pytest.raises(ValueError):
        shard(8, 2, 2)


def test_two_rank_ensemble_equals_single_process():
    import torch.multiprocessing as mp

    total, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=120) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _densities(range(total))
    assert [(r[1], r[2]) for r in results] == [(0, 3), (3, 2)]
    for rank, off, cnt, n, mean, var, gathered in results:
        assert n == total
        assert np.allclose(mean, ref.mean(axis=0), rtol=1e-13, atol=0)
        assert np.allclose(var, ref.var(axis=0), rtol=1e-9, atol=1e-15)
        assert np.array_equal(gathered, ref)          # rows ordered by global replica id, bit-identical
    assert not np.array_equal(ref[0], ref[1])          # replicas really differ (different RNG keys)
