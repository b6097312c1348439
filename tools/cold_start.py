"""How long does a GPU that has been idle take to reach its steady step time?  (what a short timed window of a process's first
seconds measures: bench.py --steps 20 on a fresh box)

One process, melbourne x 1024 (the headline workload): 20-step windows timed with HIP events, every window followed by `gap` untimed
steps, printed with the GPU-busy time accumulated so far.  Run it as the FIRST process of a gpurun call.

    python tools/cold_start.py [windows] [gap]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import replica_demand  # noqa: E402
from pednstream_amd import NetworkEnvGenerator  # noqa: E402

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")
windows = int(sys.argv[1]) if len(sys.argv) > 1 else 40
gap = int(sys.argv[2]) if len(sys.argv) > 2 else 400
R = 1024
t_proc = time.perf_counter()
net = NetworkEnvGenerator(DATA).create_network("melbourne", verbose=False, n_replicas=R, rng_seed=0)
e = net.engine()
T = net.simulation_steps
for nid in net.origin_nodes:
    net.set_demand_matrix(nid, np.stack([replica_demand(T, r) for r in range(R)]))
e.synchronize()
print(f"# set-up {time.perf_counter() - t_proc:.1f} s; windows of 20 steps (steps 101..120 of an episode), {gap} untimed steps in between")
busy_ms = 0.0
for w in range(windows):
    e.reset(lazy=True)
    e.timer_begin()
    e.run(1, 101)
    busy_ms += e.timer_end()
    e.timer_begin()
    e.run(101, 121)
    ms = e.timer_end()
    busy_ms += ms
    print(f"window {w:3d}: {ms / 20 * 1e3:6.2f} us per step   after {busy_ms:8.1f} ms of GPU work, {time.perf_counter() - t_proc:6.1f} s into the process", flush=True)
    left = gap
    t = 121
    e.timer_begin()
    while left > 0:
        k = min(left, T - t)
        e.run(t, t + k)
        left -= k
        t += k
        if t >= T:
            e.reset(lazy=True)
            t = 1
    busy_ms += e.timer_end()
net.close()
