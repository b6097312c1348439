"""Episode turnover of the batched RL env: what a reset costs next to an episode's stepping time.

    python tools/reset_time.py [network] [envs] [history]       # default 45_intersections 2048 full

reset() = histories and widths back to t = 0; options={'randomize': True} additionally draws a scenario per env, either with
the reference's randomisers env after env (mode 'reference', host-bound) or for all envs at once (mode 'vectorised').
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from pednstream_amd.rl_env import VecPedNetEnv  # noqa: E402


def main():
    network = sys.argv[1] if len(sys.argv) > 1 else "45_intersections"
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    history = sys.argv[3] if len(sys.argv) > 3 else "full"
    t0 = time.perf_counter()
    env = VecPedNetEnv(network, n_envs=B, obs_mode="option3", action_gap=1, seed=0, data_dir=os.path.join(ROOT, "data"), history=history)
    e = env.network.engine()
    e.synchronize()
    print(f"{network} x {B} envs, history {history}: construction {time.perf_counter() - t0:.3f} s")

    def timed(label, **kw):
        ts = []
        for rep in range(3):
            np.random.seed(rep)
            t0 = time.perf_counter()
            env.reset(**kw)
            e.synchronize()
            ts.append(time.perf_counter() - t0)
            # a few steps so that the next reset has something to clear
            for t in range(1, 20):
                e.rl_step(None, t, 1, fetch=False)
            e.synchronize()
        print(f"  {label:58s} {min(ts) * 1e3:9.1f} ms (best of 3: {', '.join(f'{x * 1e3:.1f}' for x in ts)})")

    timed("reset()")
    timed("reset(options={'randomize': True, 'mode': 'vectorised'})  [device]", options={"randomize": True, "mode": "vectorised"}, seed=1)
    timed("reset(options={'randomize': True, 'mode': 'vectorised_host'})  [numpy]", options={"randomize": True, "mode": "vectorised_host"}, seed=1)

    def piece(label, fn, n=5):
        ts = []
        for _ in range(n):
            e.synchronize()
            t0 = time.perf_counter()
            fn()
            e.synchronize()
            ts.append(time.perf_counter() - t0)
        print(f"    {label:56s} {min(ts) * 1e3:9.2f} ms")

    net = env.network
    origins = [nd.index for nd in net.nodes.values() if nd.virtual_incoming_link is not None and nd.node_id in net.origin_nodes]
    piece("pedn_randomize_scenarios: link parameters", lambda: e.randomize_scenarios(3, 0.2, links=True, od_weights=False))
    piece("pedn_randomize_scenarios: OD weights + tables", lambda: e.randomize_scenarios(3, 0.2, links=False, od_weights=True))
    piece(f"pedn_randomize_scenarios: demand of {len(origins)} origins", lambda: e.randomize_scenarios(3, 0.2, links=False, od_weights=False, origin_nodes=origins))
    piece("pedn_reset (histories)", lambda: e.reset())
    piece("four width matrices up (set_widths, before round 4)", lambda: [e.set_widths(code, net._widths[w]) for w, code in (("front", 0), ("back", 1), ("sep", 2), ("sepnp", 3))])
    piece("pedn_reset_widths (initial widths broadcast on the device)", lambda: e.reset_widths(net._widths["front"][:, 0], net._widths["back"][:, 0], net._widths["sep"][:, 0]))
    piece("first observation fetched", lambda: e.rl_observe(1, accumulate=False))
    timed("reset(options={'randomize': True, 'mode': 'reference'})  [reference streams]", options={"randomize": True, "mode": "reference"}, seed=1)
    T = env.simulation_steps
    t0 = time.perf_counter()
    env.reset()
    for t in range(1, T):
        e.rl_step(None, t, 1, fetch=False)
    e.synchronize()
    print(f"  one episode of {T - 1} env steps (no actions, nothing fetched)        {(time.perf_counter() - t0) * 1e3:9.1f} ms")
    env.close()


if __name__ == "__main__":
    main()
