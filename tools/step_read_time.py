"""A caller that looks at the state after every step (a controller, an output handler): network_loading(t) + two reads per step, us per
step, under the plain plan (PEDN_INLINE_TF=0 / PEDN_LINK_OWNER=0) and under the default -- pedn_step notices such a caller and steps it
under the plain plan (two launches per step instead of three).

    python tools/step_read_time.py
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from golden_util import DATA
from pednstream_amd import NetworkEnvGenerator
for name, envs in (("nine_intersections", ({"PEDN_INLINE_TF": "0"}, {})), ("od_flow_example", ({"PEDN_INLINE_TF": "0"}, {})), ("melbourne", ({"PEDN_LINK_OWNER": "0"}, {}))):
    for env in envs:
        for k in ("PEDN_INLINE_TF", "PEDN_LINK_OWNER"):
            os.environ.pop(k, None)
        os.environ.update(env)
        np.random.seed(7)
        net = NetworkEnvGenerator(DATA).create_network(name, verbose=False, n_replicas=1, rng_seed=11)
        T = net.params["simulation_steps"]
        link = next(iter(net.links.values()))
        best = 1e9
        for rep in range(3):
            net.reset()
            net.engine().synchronize()
            t0 = time.perf_counter()
            acc = 0.0
            for t in range(1, T):
                net.network_loading(t)
                acc += float(link.density[t]) + float(link.inflow[t])       # a controller looks at the state after every step
            net.engine().synchronize()
            best = min(best, time.perf_counter() - t0)
        print(f"{name} {env or 'default'}: step + two reads {best / (T - 1) * 1e6:.1f} us per step (acc {acc:.3f})", flush=True)
        net.close()
