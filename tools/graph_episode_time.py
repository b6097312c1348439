#!/usr/bin/env python3
"""Whole episodes of the graph-replayed rollout (45_intersections x 2048 envs, 3-layer MLP policy, 4 policy steps per graph) WITH their
device-randomised resets: where an episode's time goes, and what a host synchronisation at different places changes.

    python tools/graph_episode_time.py

Found with it (profiles/EXPERIMENTS.md #69): an eager step enqueued behind replays that are still in flight -- policy kernels plus a
cross-stream wait on the engine's stream -- slowed every dispatch of the replays (174 replays 44.7 -> 51.5 ms); GraphedRollout.step now
waits for its stream before it enqueues anything eager, and the variants below agree."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pednstream_amd.rl_env import VecPedNetEnv
B = 2048
def make():
    env = VecPedNetEnv("45_intersections", n_envs=B, obs_mode="option3", action_gap=1, seed=0, data_dir=os.path.join(ROOT, "data"), history="recent")
    low = torch.as_tensor(env.action_low, device="cuda", dtype=torch.float64)
    span = torch.as_tensor(env.action_high, device="cuda", dtype=torch.float64) - low
    torch.manual_seed(0)
    mlp = torch.nn.Sequential(torch.nn.Linear(env.n_obs, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(), torch.nn.Linear(64, env.n_actions), torch.nn.Sigmoid()).to("cuda").requires_grad_(False)
    ret = torch.zeros(B, device="cuda")
    policy = lambda obs: (low + span * mlp(obs).double()).contiguous()
    return env, env.capture(policy, lambda o, r: ret.add_(r[:, 0]), steps_per_replay=4)
for variant in ("A: sync before the tail", "B: no sync anywhere", "C: sync after reset only", "D: sync after reset, eager step and before the tail"):
    env, roll = make()
    for ep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.reset(options={"randomize": True, "mode": "vectorised"}, seed=50 + ep)
        if variant[0] in "CD":
            torch.cuda.synchronize()
        roll.step()
        if variant[0] == "D":
            torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        while env.sim_step + 4 <= env.simulation_steps:
            roll.step()
        e1.record()
        if variant[0] in "AD":
            torch.cuda.synchronize()
        t3 = time.perf_counter()
        while not roll.step():
            pass
        torch.cuda.synchronize()
        t5 = time.perf_counter()
    print(f"{variant}: burst on the device {e0.elapsed_time(e1):.2f} ms, tail {1e3*(t5-t3):.2f} ms, episode {1e3*(t5-t0):.2f} ms", flush=True)
    env.close()
