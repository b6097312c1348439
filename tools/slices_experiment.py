"""Throughput of S independent engines (replica slices, one HIP stream each) sharing ONE GPU.

Replicas are independent, so a batch of R replicas can be run as S engines of R/S replicas with global replica ids
``[s*R/S, (s+1)*R/S)`` (the same results, see tests/test_gpu_parity.py::test_replica_offset_blocks).  The launches of one
engine serialise on its stream; with S > 1 the fill/drain of one slice's launch overlaps with the other slices' kernels.

    python tools/slices_experiment.py --network melbourne --replicas 1024 --slices 1 2 4
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bench import replica_demand  # noqa: E402
from pednstream_amd import NetworkEnvGenerator  # noqa: E402


def build(network, n, offset):
    gen = NetworkEnvGenerator(os.path.join(ROOT, "data"))
    net = gen.create_network(network, verbose=False, n_replicas=n, replica_offset=offset, rng_seed=0, device=0)
    e = net.engine()
    for nid in net.origin_nodes:
        e.set_demand_matrix(net.nodes[nid].index, np.stack([replica_demand(net.simulation_steps, offset + r) for r in range(n)]))
    e.synchronize()
    return net, e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--network", default="melbourne")
    ap.add_argument("--replicas", type=int, default=1024)
    ap.add_argument("--slices", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--chunk", type=int, default=10, help="steps enqueued per engine before moving to the next one")
    args = ap.parse_args()
    for S in args.slices:
        n = args.replicas // S
        pairs = [build(args.network, n, s * n) for s in range(S)]
        L = pairs[0][1].n_links

        def advance(t0, k):
            t = t0
            while t < t0 + k:
                c = min(args.chunk, t0 + k - t)
                for _, e in pairs:
                    e.run(t, t + c)
                t += c
            return t

        t = advance(1, args.warmup)
        for _, e in pairs:
            e.synchronize()
        w0 = time.perf_counter()
        advance(t, args.steps)
        for _, e in pairs:
            e.synchronize()
        wall = time.perf_counter() - w0
        sums = [float(np.asarray(net.read_field("cumulative_inflow", t + args.steps - 1, t + args.steps)).sum()) for net, _ in pairs]
        print(json.dumps({"network": args.network, "replicas": args.replicas, "slices": S, "us_per_step": wall / args.steps * 1e6,
                          "link_updates_per_s": L * n * S * args.steps / wall, "checksum": sum(sums)}), flush=True)
        for net, _ in pairs:
            net.close()


if __name__ == "__main__":
    main()
