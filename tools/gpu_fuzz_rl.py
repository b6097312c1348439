"""GPU-vs-restatement fuzz of the batched RL step: random scenarios from tests/fuzz_cases.py that contain controller nodes
or separator links, random observation mode / normalisation / action gap, random actions (also outside the bounds); the
observations and rewards of VecPedNetEnv must equal tests/rl_oracle.py (the reference's RL glue restated on the C oracle,
itself pinned by the rl_* goldens) bit for bit.

    python tools/gpu_fuzz_rl.py 4000 4400          # seeds; PEDN_FUSE_OBS=0 keeps the observations in their own launch
    python tools/gpu_fuzz_rl.py 4000 4400 clocked  # every step but the first through the device-resident step clock
                                                   # (pedn_rl_clock_begin / pedn_rl_step_clocked: the launches a captured graph replays)
"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from fuzz_cases import random_case  # noqa: E402
from pednstream_amd import Network  # noqa: E402
from pednstream_amd.flatten import flatten_network  # noqa: E402
from pednstream_amd.rl_env import VecPedNetEnv  # noqa: E402
from rl_oracle import RlOracle  # noqa: E402
from test_rl_golden import agent_spec  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
clocked = len(sys.argv) > 3 and sys.argv[3] == "clocked"
if clocked:
    import torch
B, steps = 4, 30
ran = no_agents = skipped = 0
for seed in range(lo, hi):
    if seed > lo and (seed - lo) % 100 == 0:     # heartbeat (gpurun kills a run that is silent for 7 minutes)
        print(f"#   ... seed {seed} of {lo}..{hi}", flush=True)
    adj, params, origins, dests = random_case(seed)
    np.random.seed(seed)
    try:
        net = Network(adj, copy.deepcopy(params), origin_nodes=origins, destination_nodes=dests, verbose=False, n_replicas=B, rng_seed=seed)
    except KeyError:
        skipped += 1
        continue
    if flatten_network(net)["max_degree"] > 8:   # kernel limit (PEDN_MAX_DEGREE): pedn_create refuses such a junction
        skipped += 1
        net.close()
        continue
    rng = np.random.default_rng(seed)
    mode = ["option1", "option2", "option3", "option4", "option5"][rng.integers(0, 5)]
    normalize, gap = bool(rng.integers(0, 2)), int(rng.integers(1, 3))
    try:
        env = VecPedNetEnv("fuzz", n_envs=B, obs_mode=mode, normalize_obs=normalize, action_gap=gap, network=net, reward_mode="all")
    except ValueError:
        no_agents += 1
        net.close()
        continue
    except IndexError:                      # option4 + normalisation indexes past the feature row in the reference too
        skipped += 1
        net.close()
        continue
    spec = agent_spec(net)
    model = flatten_network(net)
    checks = {r: RlOracle(net, model, spec, mode, normalize, gap, seed=seed, replica=r, reward_mode="all") for r in (0, B - 1)}
    n = min(steps, (net.simulation_steps - 1) // gap)
    hi_w = float(max(l.width for l in net.links.values())) + 0.5
    eng = net.engine()
    for k in range(n):
        acts = rng.uniform(-0.5, hi_w, size=(B, env.n_actions)).astype(np.float32)
        if clocked and k > 0:
            if not eng.rl_clocked():
                eng.rl_clock_begin(env.sim_step)
            a = torch.as_tensor(acts.astype(np.float64), device="cuda")
            torch.cuda.synchronize()
            eng.rl_step_clocked(a.data_ptr(), gap)
            torch.cuda.synchronize()
            env.sim_step += gap
            obs, rew = (x.cpu().numpy() for x in env.device_views())
            if k % 11 == 10:                  # now and then something else touches the engine: the clocked section ends and begins again
                assert eng.rl_clock_end() == env.sim_step
        else:
            obs, rew, *_ = env.step(acts)
        for r, orc in checks.items():
            o, w = orc.step(acts[r])
            assert np.array_equal(obs[r], o), (seed, k, r, mode, normalize, gap)
            assert np.array_equal(rew[r], w), (seed, k, r, mode, normalize, gap)
    ran += 1
    env.close()
print(f"{'clocked steps, ' if clocked else ''}fuse_obs={os.environ.get('PEDN_FUSE_OBS', 'auto')} seeds {lo}..{hi}: {ran} controlled scenarios x 2 checked replicas bit-exact on observations and "
      f"rewards, {no_agents} scenarios without agents, {skipped} skipped (the reference raises there too)")
