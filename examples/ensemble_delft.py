"""A Monte-Carlo ensemble of the Delft network on one MI355X: 1024 replicas of the same scenario, each with its own random
numbers and its own Poisson demand series, stepped together (the two halves of the batch as two chains of launches, see
DESIGN.md section 5), then the ensemble mean and spread of every link's density -- the workload the engine is built for
(BASELINE config #3).  Under `python -m torch.distributed.run --nproc-per-node N` every rank owns 1024 replicas of a
larger ensemble and the moments are reduced over all of them (RCCL).

    python examples/ensemble_delft.py [n_replicas]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pednstream_amd import NetworkEnvGenerator, ensemble  # noqa: E402
from pednstream_amd.network import LINK_FIELDS  # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl")
    offset = rank * R                                   # global replica ids key the RNG and the demand: any rank count gives the same ensemble
    net = NetworkEnvGenerator(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")).create_network(
        "delft", verbose=False, n_replicas=R, replica_offset=offset, rng_seed=0, device=int(os.environ.get("LOCAL_RANK", "0")))
    T = net.simulation_steps
    base = {nid: np.asarray(net.nodes[nid].demand, dtype=np.float64)[:T] for nid in net.origin_nodes}
    for nid in net.origin_nodes:                        # every replica: a Poisson realisation of the scenario's demand profile
        net.set_demand_matrix(nid, np.stack([np.random.default_rng(offset + r).poisson(np.maximum(base[nid], 0.0)).astype(np.float64)
                                             for r in range(R)]))
    e = net.engine()
    e.synchronize()
    t0 = time.perf_counter()
    net.run(1, T)                                       # all 499 steps of all replicas
    e.synchronize()
    dt = time.perf_counter() - t0
    dens = e.read_block(LINK_FIELDS["density"][0], T - 1, T)[0]          # [links, replicas] at the last step
    n, mean, var = ensemble.ensemble_moments(dens.T)                    # over ALL ranks' replicas
    if rank == 0:
        worst = int(np.argmax(mean))
        link = net._link_list[worst]
        print(f"{n} replicas x {T - 1} steps of delft in {dt * 1e3:.1f} ms on this rank ({e.n_links * R * (T - 1) / dt:.3g} link-updates/s)")
        print(f"densest link at the end: {link.link_id}  mean {mean[worst]:.3f} ped/m2, standard deviation over the ensemble {np.sqrt(var[worst]):.3f}")
        print(f"links above 2 ped/m2 on average: {int((mean > 2.0).sum())} of {e.n_links}; share of replicas in which the densest link is above "
              f"its critical density: {float((dens[worst] > link.k_critical).mean()):.2f}")
    net.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
