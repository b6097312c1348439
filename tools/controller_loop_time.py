"""A host-side controller in the loop (the reference's rule-based gaters, examples/forky_queues.py): per step network_loading(t), ONE gate
width set through the link view and ONE density read, single replica: us per step.

    python tools/controller_loop_time.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pednstream_amd import NetworkEnvGenerator  # noqa: E402

for name in ("od_flow_example", "nine_intersections", "melbourne"):
    np.random.seed(7)
    net = NetworkEnvGenerator(os.path.join(ROOT, "data")).create_network(name, verbose=False, n_replicas=1, rng_seed=11)
    T = net.params["simulation_steps"]
    link = next(iter(net.links.values()))
    w0 = link.width
    out = {}
    for label in ("step + set", "step + set + read"):
        best = 1e9
        for rep in range(3):
            net.reset()
            net.engine().synchronize()
            t0 = time.perf_counter()
            acc = 0.0
            for t in range(1, T):
                net.network_loading(t)
                link.back_gate_width = w0 * (0.5 + 0.5 * ((t * 7) % 10) / 10.0)
                if label.endswith("read"):
                    acc += float(link.density[t])
            net.engine().synchronize()
            best = min(best, time.perf_counter() - t0)
        out[label] = best / (T - 1) * 1e6
    print(f"{name}: " + ", ".join(f"{k} {v:.1f} us per step" for k, v in out.items()), flush=True)
    net.close()
