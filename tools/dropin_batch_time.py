"""The reference's calling sequence -- `for t in range(1, T): network.network_loading(t)` -- on a BATCH of replicas, against run(1, T):
what does the per-step call cost when the launches are long?  (tools/dropin_time.py is the single-replica case)

    python tools/dropin_batch_time.py [replicas] [networks ...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import replica_demand  # noqa: E402
from pednstream_amd import NetworkEnvGenerator  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for name in sys.argv[2:] or ("melbourne", "delft"):
    net = NetworkEnvGenerator(os.path.join(ROOT, "data")).create_network(name, verbose=False, n_replicas=R, rng_seed=0)
    e = net.engine()
    T = net.simulation_steps
    for nid in net.origin_nodes:
        net.set_demand_matrix(nid, np.stack([replica_demand(T, r) for r in range(R)]))
    out = {}
    for how in ("run", "loop", "run", "loop"):
        e.reset()
        e.synchronize()
        t0 = time.perf_counter()
        if how == "run":
            net.run(1, T)
        else:
            for t in range(1, T):
                net.network_loading(t)
        net.synchronize()
        out.setdefault(how, []).append((time.perf_counter() - t0) / (T - 1) * 1e6)
    print(f"{name} x {R}: run(1, T) {min(out['run']):.2f} us per step, for t: network_loading(t) {min(out['loop']):.2f} us per step "
          f"(plan: {e.plan_info()['chains']} chains)", flush=True)
    net.close()
