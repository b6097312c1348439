"""What does a 20-step timed region (the driver's bench command) consist of?  Per window: host time to ENQUEUE the 20 steps (pedn_run
returning), wall time to completion, device time by HIP events -- the first windows of a process, back to back.

    python tools/short_window.py [windows]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import replica_demand  # noqa: E402
from pednstream_amd import NetworkEnvGenerator  # noqa: E402

windows = int(sys.argv[1]) if len(sys.argv) > 1 else 12
R = 1024
net = NetworkEnvGenerator(os.path.join(ROOT, "data")).create_network("melbourne", verbose=False, n_replicas=R, rng_seed=0)
e = net.engine()
T = net.simulation_steps
for nid in net.origin_nodes:
    net.set_demand_matrix(nid, np.stack([replica_demand(T, r) for r in range(R)]))
e.synchronize()
e.run(1, 6)
e.synchronize()
t = 6
for w in range(windows):
    if t + 20 >= T:
        e.reset(lazy=True)
        e.run(1, 6)
        e.synchronize()
        t = 6
    e.timer_begin()
    t0 = time.perf_counter()
    e.run(t, t + 20)
    t1 = time.perf_counter()
    dev = e.timer_end()
    t2 = time.perf_counter()
    print(f"window {w:2d} (steps {t}..{t + 19}): enqueue {1e6 * (t1 - t0) / 20:5.1f} us per step of host time, wall {1e6 * (t2 - t0) / 20:5.1f}, device {dev * 1e3 / 20:5.1f}", flush=True)
    t += 20
net.close()
