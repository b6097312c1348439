#!/usr/bin/env python3
"""What bounds the graph-replayed rollout of config #5 (45_intersections x 2048 envs)?

    python tools/graph_rollout_time.py [n_envs] [policy: none|random|mlp] [steps_per_replay ...]

Per configuration: us per policy step end to end, us of HOST time per roll.step() call (the replay's enqueue), replays / eager steps.
policy none = constant actions (no policy kernels, no on_step): the env step's two launches alone in a graph.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations inside the replayed graph."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pednstream_amd.rl_env import VecPedNetEnv  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
which = sys.argv[2] if len(sys.argv) > 2 else "all"
spr = [int(x) for x in sys.argv[3:]] or [1, 4, 16]
env = VecPedNetEnv("45_intersections", n_envs=B, obs_mode="option3", action_gap=1, seed=0, data_dir=os.path.join(ROOT, "data"), history="recent")
low = torch.as_tensor(env.action_low, device="cuda", dtype=torch.float64)
span = torch.as_tensor(env.action_high, device="cuda", dtype=torch.float64) - low
gen = torch.Generator(device="cuda").manual_seed(0)
torch.manual_seed(0)
mlp = torch.nn.Sequential(torch.nn.Linear(env.n_obs, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                          torch.nn.Linear(64, env.n_actions), torch.nn.Sigmoid()).to("cuda").requires_grad_(False)
const = (low + 0.5 * span).expand(B, env.n_actions).contiguous()
ret = torch.zeros(B, device="cuda")
policies = {"none": (lambda obs: const, None),
            "random": (lambda obs: low + span * torch.rand((B, env.n_actions), generator=gen, device="cuda", dtype=torch.float64), lambda o, r: ret.add_(r[:, 0])),
            "mlp": (lambda obs: (low + span * mlp(obs).double()).contiguous(), lambda o, r: ret.add_(r[:, 0]))}
for name, (policy, on_step) in policies.items():
    if which not in ("all", name):
        continue
    for n in spr:
        env.reset(options={"randomize": True, "mode": "vectorised"} if os.environ.get("GRT_RANDOMIZE") else None, seed=3)
        roll = env.capture(policy, on_step, generators=[gen], steps_per_replay=n)
        for _ in range(4):
            roll.step()
        torch.cuda.synchronize()
        s0, host, calls = env.sim_step, 0.0, 0
        t0 = time.perf_counter()
        while env.sim_step + n <= env.simulation_steps - 2 and calls < 640 // n:
            h0 = time.perf_counter()
            roll.step()
            host += time.perf_counter() - h0
            calls += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = env.sim_step - s0
        print(f"{B} envs{' (per-env scenarios)' if os.environ.get('GRT_RANDOMIZE') else ''}{' CLK_CHAINS=' + os.environ['PEDN_CLK_CHAINS'] if 'PEDN_CLK_CHAINS' in os.environ else ''}, policy {name:6s}, {n:2d} policy steps per replay: {dt / steps * 1e6:6.1f} us per policy step = {B * steps / dt:.3e} env-steps/s; "
              f"host {host / calls * 1e6:6.1f} us per replay call; replays {roll.replays}, eager {roll.eager_steps}", flush=True)
env.close()
