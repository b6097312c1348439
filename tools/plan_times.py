"""Device time per step of pedn_run under the launch plan the environment selects, for A/B runs inside ONE gpurun call (boxes differ
by +-1 us per kernel): HIP-event time of 300 steps after 100, and the per-launch durations pedn_profile_run reports for that plan.

    PEDN_LINK_OWNER=1 python tools/plan_times.py melbourne delft
    PEDN_HIP_LIB=$PWD/pednstream_amd/csrc/libpedn_hip_base.so python tools/plan_times.py melbourne:2048
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import replica_demand  # noqa: E402
from pednstream_amd import NetworkEnvGenerator  # noqa: E402

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")
KEYS = ("PEDN_LINK_OWNER", "PEDN_INLINE_TF", "PEDN_FUSE_TP", "PEDN_STREAMS", "PEDN_PACK_BY_LOAD", "PLAN_DEMAND_SCALE")
tag = " ".join(f"{k[5:]}={os.environ[k]}" for k in KEYS if k in os.environ)
lib = os.path.basename(os.environ.get("PEDN_HIP_LIB", "libpedn_hip.so"))
for spec in sys.argv[1:]:
    network, _, reps = spec.partition(":")
    R = int(reps or 1024)
    net = NetworkEnvGenerator(DATA).create_network(network, verbose=False, n_replicas=R, rng_seed=0,
                                                   history=os.environ.get("PLAN_HISTORY", "full"))
    e = net.engine()
    T = net.simulation_steps
    n = min(300, T - 142)
    for nid in net.origin_nodes:
        sc = float(os.environ.get("PLAN_DEMAND_SCALE", "1"))          # 12: the congested variant of the bench's demand
        net.set_demand_matrix(nid, np.stack([replica_demand(T, r, base=5.0 * sc, peak=10.0 * sc) for r in range(R)]))
    e.run(1, 101)
    e.synchronize()
    best = 1e9
    for _ in range(3):
        e.reset()
        e.run(1, 101)
        e.synchronize()
        e.timer_begin()
        e.run(101, 101 + n)
        best = min(best, e.timer_end() / n * 1e3)
    plan = e.plan_info()
    (tf, node, link), chains = e.profile_run(101 + n, 101 + n + 40)
    rc, _ = e.error_flags()
    print(f"{network:10s} x{R:5d} {lib:22s} {tag or 'default plan':28s} {best:6.2f} us/step   chains {chains}  node_kernel {node * 1e3:6.2f}  "
          f"second launch {link * 1e3:6.2f}  stand-alone tf {tf * 1e3:5.2f}  flags {rc}  packed by {plan['packed_by']}  probe {plan['stream_probe_attempts']}x {plan['stream_probe_us']} us", flush=True)
    net.close()
