"""Random-policy rollout of 2048 parallel 45_intersections environments on one GPU (BASELINE config #5).

    python examples/vec_env_rollout.py [n_envs] [steps]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pednstream_amd.rl_env import VecPedNetEnv  # noqa: E402


def main():
    n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    env = VecPedNetEnv("45_intersections", n_envs=n_envs, obs_mode="option3", action_gap=1, seed=0)
    print("agents", env.possible_agents, "action dims", env.n_actions, "obs dims", env.n_obs)
    obs, _ = env.reset(options={"randomize": True}, seed=1)
    rng = np.random.default_rng(0)
    ret = np.zeros(n_envs)
    t0 = time.perf_counter()
    for _ in range(steps):
        actions = rng.uniform(env.action_low, env.action_high, size=(n_envs, env.n_actions))
        obs, rew, terminated, truncated, info = env.step(actions)
        ret += rew[:, 0]
    dt = time.perf_counter() - t0
    print(f"{n_envs * steps / dt:.3g} env-steps/s including host-side action sampling and obs/reward copies; "
          f"mean return of the first agent {ret.mean():.1f}")
    env.close()


if __name__ == "__main__":
    main()
