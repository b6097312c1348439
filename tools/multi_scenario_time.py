"""MultiScenarioVecEnv (one engine per randomised TOPOLOGY, the reference's randomize_network with its moved OD nodes): us per vector step.

    python tools/multi_scenario_time.py [n_envs] [group_size]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pednstream_amd.rl_env import MultiScenarioVecEnv  # noqa: E402

n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
group = int(sys.argv[2]) if len(sys.argv) > 2 else 64
t0 = time.perf_counter()
env = MultiScenarioVecEnv("45_intersections", n_envs=n_envs, group_size=group, obs_mode="option3", data_dir=os.path.join(ROOT, "data"), seed=0,
                          history="recent") if "history" in MultiScenarioVecEnv.__init__.__code__.co_varnames else \
    MultiScenarioVecEnv("45_intersections", n_envs=n_envs, group_size=group, obs_mode="option3", data_dir=os.path.join(ROOT, "data"), seed=0)
t1 = time.perf_counter()
env.reset(options={"randomize": True}, seed=1)
t2 = time.perf_counter()
a = np.tile(((env.action_low + env.action_high) / 2)[None], (n_envs, 1)).astype(np.float64)
serial = len(sys.argv) > 3 and sys.argv[3] == "serial"     # the groups one after the other, each step waited for (how step() worked before)


def step():
    if not serial:
        return env.step(a)
    off = 0
    for g, size in zip(env.groups, env.sizes):
        g.step(a[off:off + size])
        off += size


for _ in range(20):
    step()
n = 200
t3 = time.perf_counter()
for _ in range(n):
    step()
dt = (time.perf_counter() - t3) / n
print(f"{n_envs} envs in {len(env.groups)} groups of {group} (one engine each): construction {t1 - t0:.2f} s, randomised reset (new topologies) {t2 - t1:.2f} s, "
      f"{'groups stepped one after the other: ' if serial else ''}{dt * 1e6:.0f} us per vector step = {n_envs / dt:.3g} env-steps/s", flush=True)
env.close()
