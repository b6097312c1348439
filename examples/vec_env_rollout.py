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

    # the same rollout with the policy on the GPU: actions are sampled by torch on the device, observations and rewards are
    # torch views of the engine's buffers -- nothing crosses PCIe (torch may be imported after the engine: one HIP runtime)
    try:
        import torch
    except ImportError:
        env.close()
        return
    env.reset(options={"randomize": True, "mode": "vectorised"}, seed=2)
    gen = torch.Generator(device="cuda").manual_seed(0)
    low = torch.as_tensor(env.action_low, device="cuda", dtype=torch.float64)
    span = torch.as_tensor(env.action_high, device="cuda", dtype=torch.float64) - low
    ret_d = torch.zeros(n_envs, device="cuda")
    for _ in range(20):                                   # torch's first-use initialisation is not part of either rate below
        (low + span * torch.rand((n_envs, env.n_actions), generator=gen, device="cuda", dtype=torch.float64)).sum()
    gen.manual_seed(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        actions = low + span * torch.rand((n_envs, env.n_actions), generator=gen, device="cuda", dtype=torch.float64)
        obs, rew, terminated = env.step_device(actions)
        ret_d += rew[:, 0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n_envs * steps / dt:.3g} env-steps/s with the random policy on the GPU (step_device); "
          f"mean return of the first agent {float(ret_d.mean()):.1f}")

    # and without any host synchronisation: policy kernels, env step and the consumer of the observations chained by events
    env.reset(options={"randomize": True, "mode": "vectorised"}, seed=2)
    gen = torch.Generator(device="cuda").manual_seed(0)
    ret_a = torch.zeros(n_envs, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        actions = low + span * torch.rand((n_envs, env.n_actions), generator=gen, device="cuda", dtype=torch.float64)
        obs, rew, terminated = env.step_device(actions, sync=False)
        ret_a += rew[:, 0]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n_envs * steps / dt:.3g} env-steps/s with step_device(sync=False): the host never waits; "
          f"mean return of the first agent {float(ret_a.mean()):.1f} (the same rollout)")

    # and without the host in the loop at all: (policy -> env step -> reward sum) captured ONCE as a torch.cuda.CUDAGraph and replayed --
    # the env step's launches have constant arguments because its step index lives in device memory (include/pedn.h: pedn_rl_step_clocked)
    env.reset(options={"randomize": True, "mode": "vectorised"}, seed=2)
    gen = torch.Generator(device="cuda").manual_seed(0)
    ret_g = torch.zeros(n_envs, device="cuda")
    roll = env.capture(lambda obs: low + span * torch.rand((n_envs, env.n_actions), generator=gen, device="cuda", dtype=torch.float64),
                       lambda obs, rew: ret_g.add_(rew[:, 0]), generators=[gen], steps_per_replay=4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while env.sim_step <= steps:
        roll.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n_envs * (env.sim_step - 1) / dt:.3g} env-steps/s graph-replayed ({roll.replays} replays of 4 policy steps, {roll.eager_steps} eager steps); "
          f"mean return of the first agent {float(ret_g.mean()):.1f}")
    env.close()


if __name__ == "__main__":
    main()
