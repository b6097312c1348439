"""Host-facing batched env step (numpy actions in, observations / rewards out, every call synchronous): us per call by batch size --
the path an SB3-style learner with a host policy drives (VecPedNetEnv.step / PedNetSB3VecEnv.step_wait).

    python tools/host_step_time.py [n_envs ...]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pednstream_amd.rl_env import VecPedNetEnv  # noqa: E402

for B in [int(x) for x in sys.argv[1:]] or [1, 64, 2048]:
    env = VecPedNetEnv("45_intersections", n_envs=B, obs_mode="option3", action_gap=1, seed=0, data_dir=os.path.join(ROOT, "data"), history="recent")
    a = np.tile(((env.action_low + env.action_high) / 2)[None], (B, 1)).astype(np.float64)
    env.reset()
    for _ in range(50):
        env.step(a)
    best = {}
    for label, kw in (("step(actions)", {}), ("step(actions, fetch=False)", {"fetch": False}), ("step(None)", None)):
        env.reset()
        for _ in range(20):
            env.step(a)
        t0 = time.perf_counter()
        n = 400
        for _ in range(n):
            if kw is None:
                env.step(None)
            else:
                env.step(a, **kw)
        env.network.synchronize()
        best[label] = (time.perf_counter() - t0) / n * 1e6
    print(f"{B:5d} envs: " + ", ".join(f"{k} {v:7.1f} us" for k, v in best.items()) + f"  -> {B / best['step(actions)'] * 1e6:.3g} env-steps/s through the host path", flush=True)
    env.close()
