"""GPU tests of the batched RL step (config #5 caller): device action clipping / observation / reward kernels against the
goldens captured from the reference's rl modules and against the CPU restatement for many envs."""
import numpy as np
import pytest

from golden_util import Golden, build_network, compare_fields
from pednstream_amd.flatten import flatten_network
from pednstream_amd.network import LINK_FIELDS
from pednstream_amd.rl_env import PedNetParallelEnv, VecPedNetEnv
from rl_oracle import RlOracle
from test_rl_golden import RL_CASES

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", RL_CASES)
def test_vec_env_reproduces_reference_rl_goldens(case):
    g = Golden(case)
    rl = g.info["rl"]
    net = build_network(g, n_replicas=1, replica_offset=g.replica, rng_seed=g.seed)
    env = VecPedNetEnv(g.info["scenario"], n_envs=1, obs_mode=rl["obs_mode"], normalize_obs=rl["normalize"],
                       action_gap=rl["action_gap"], network=net)
    assert env.possible_agents == [a["id"] for a in rl["agents"]]
    acts, ref_obs, ref_rew = g.state("rl_actions"), g.state("rl_obs"), g.state("rl_rewards")
    assert (env.n_actions, env.n_obs) == (acts.shape[1], ref_obs.shape[1])
    for k in range(rl["env_steps"]):
        obs, rew, term, trunc, _ = env.step(acts[k:k + 1])
        assert np.array_equal(obs[0], ref_obs[k]), (k, obs[0], ref_obs[k])
        assert np.array_equal(rew[0], ref_rew[k]), (k, rew[0], ref_rew[k])
        assert term is False
    e = net._engine
    problems = compare_fields(lambda name: e.read_block(LINK_FIELDS[name][0], 0, g.steps)[:, :, 0].T, g, e.n_links, g.steps)
    assert not problems, "\n".join(problems)
    # the host mirror of the widths follows the device-side actions
    first = rl["agents"][0]
    u, v = (int(x) for x in first["links"][0].split("_"))
    w = net.links[(u, v)].separator_width if first["type"] == "sep" else net.links[(u, v)].back_gate_width
    assert 0.0 <= w <= net.links[(u, v)].width
    env.close()


def test_many_envs_match_cpu_restatement():
    g = Golden("rl_nine_opt3")
    B, steps = 64, 50
    net = build_network(g, n_replicas=B, rng_seed=4)
    env = VecPedNetEnv("nine_intersections", n_envs=B, obs_mode="option5", action_gap=2, network=net, reward_mode="all")
    rng = np.random.default_rng(9)
    actions = rng.uniform(-0.5, 4.5, size=(steps, B, env.n_actions)).astype(np.float32)
    model = flatten_network(net)
    checks = {r: RlOracle(net, model, g.info["rl"]["agents"], "option5", False, 2, seed=4, replica=r, reward_mode="all") for r in (0, 17, 63)}
    for k in range(steps):
        obs, rew, *_ = env.step(actions[k])
        for r, orc in checks.items():
            o, w = orc.step(actions[k, r])
            assert np.array_equal(obs[r], o), (k, r)
            assert np.array_equal(rew[r], w), (k, r)
    assert (rew[:, 1:] != 0).any()          # reward_mode="all": agents after the first are rewarded too
    env.close()


def test_single_env_facade_and_reset(tmp_path):
    g = Golden("rl_nine_opt3")
    env = PedNetParallelEnv("nine_intersections", obs_mode="option3", seed=g.seed, network=build_network(g, n_replicas=1, replica_offset=g.replica, rng_seed=g.seed))
    acts, ref_obs, ref_rew = g.state("rl_actions"), g.state("rl_obs"), g.state("rl_rewards")

    def run(n):
        out = []
        for k in range(n):
            actions = {a: acts[k, sl] for a, sl in env._vec.action_slices.items()}
            obs, rew, term, trunc, info = env.step(actions)
            flat = np.concatenate([obs[a] for a in env.possible_agents])
            assert np.array_equal(flat, ref_obs[k]) and np.array_equal(np.float32([rew[a] for a in env.possible_agents]), ref_rew[k])
            assert set(obs) == set(env.possible_agents) and not any(term.values())
            assert all(info[a]["step"] == env.sim_step for a in env.possible_agents)          # pz_pednet_env.py:631-642
            out.append(flat)
        return out

    obs0, _ = env.reset()
    for a in env.possible_agents:
        sp, ob = env.action_space(a), env.observation_space(a)
        assert sp.shape == (len(env.agent_manager.get_gater_outgoing_links(a)),) and (sp.low == 0).all() and (sp.high == 4).all()
        assert ob.shape == obs0[a].shape and sp.contains(sp.sample())
    assert all((v == np.float32([0, 0, 0, 0, 4] * (len(v) // 5))).all() for v in obs0.values())   # initial obs [0,0,0,0,width]
    first = run(30)
    env.reset()
    second = run(30)                                  # reset restores state and widths: identical replay
    assert all(np.array_equal(a, b) for a, b in zip(first, second))
    with pytest.raises(ValueError):
        env.step({"gate_99": np.zeros(3)})
    assert env.render() is None
    total = {a: 0.0 for a in env.possible_agents}
    env.reset()
    for k in range(5):
        _, rew, _, _, info = env.step({a: acts[k, sl] for a, sl in env._vec.action_slices.items()})
        for a in total:
            total[a] += rew[a]
    assert all(info[a]["cumulative_reward"] == total[a] for a in total)
    env.save("facade_save", base_dir=str(tmp_path))
    assert sorted(p.name for p in (tmp_path / "facade_save").iterdir()) == ["link_data.json", "network_params.json", "node_data.json"]
    env.close()


def test_history_reads_after_reset_are_not_served_from_the_previous_episode():
    """Two episodes with different actions ending at the same step: the cached column read (link.density[...]) must equal a
    fresh read_block in both (the cache key carries an epoch that every reset / demand / parameter upload bumps)."""
    g = Golden("rl_nine_opt3")
    net = build_network(g, n_replicas=2, rng_seed=g.seed)
    env = VecPedNetEnv("nine_intersections", n_envs=2, obs_mode="option3", network=net)
    e = net.engine()
    first = g.info["rl"]["agents"][0]
    lk = net.links[tuple(int(x) for x in first["links"][0].split("_"))]
    seen = []
    for episode, width in enumerate((0.2, 3.8)):
        env.reset()
        for k in range(120):
            env.step(np.full((2, env.n_actions), width))
        col = np.asarray(lk.density)[:121].copy()                      # column read through the cache
        fresh = e.read_block(LINK_FIELDS["density"][0], 0, 121, lk.index, lk.index + 1, 0, 1).reshape(-1)
        assert np.array_equal(col, fresh), episode
        assert lk.inflow[100] == e.read_element(LINK_FIELDS["inflow"][0], lk.index, 0, 100)     # element read agrees too
        seen.append(col)
    assert not np.array_equal(seen[0], seen[1])                        # the two episodes really differ
    env.close()


@pytest.mark.parametrize("case", ["rl_nine_partial", "rl_corridor_partial"])
def test_partial_action_dicts_leave_the_other_agents_alone(case):
    """PettingZoo facade with action dicts that miss agents (NaN rows of the golden): apply_all_actions only touches the agents
    it is given (rl/builders.py:343-352) -- a missing separator agent keeps a width outside the clip band and its float32
    density division."""
    g = Golden(case)
    rl = g.info["rl"]
    env = PedNetParallelEnv(g.info["scenario"], obs_mode=rl["obs_mode"], seed=g.seed,
                            network=build_network(g, n_replicas=1, replica_offset=g.replica, rng_seed=g.seed))
    acts, ref_obs, ref_rew = g.state("rl_actions"), g.state("rl_obs"), g.state("rl_rewards")
    assert np.isnan(acts).any() and not np.isnan(acts).all()
    for k in range(rl["env_steps"]):
        actions = {a: acts[k, sl] for a, sl in env._vec.action_slices.items() if not np.isnan(acts[k, sl]).any()}
        obs, rew, term, trunc, info = env.step(actions)
        flat = np.concatenate([obs[a] for a in env.possible_agents])
        assert np.array_equal(flat, ref_obs[k]), k
        assert np.array_equal(np.float32([rew[a] for a in env.possible_agents]), ref_rew[k]), k
    env.close()


def test_reference_index_errors_are_reported():
    g = Golden("rl_nine_opt4")
    with pytest.raises(IndexError):
        VecPedNetEnv("nine_intersections", n_envs=1, obs_mode="option4", normalize_obs=True, network=build_network(g))
    with pytest.raises(ValueError):
        VecPedNetEnv("nine_intersections", n_envs=1, obs_mode="option9", network=build_network(g))


def _step_device_check():
    """step_device (torch CUDA actions in, torch views of the engine's observation / reward buffers out) == step."""
    import torch

    assert torch.cuda.is_available(), "torch sees no GPU: two HIP runtimes in the process?"
    g = Golden("rl_i45_opt3")
    B, steps = 64, 25
    envs = [VecPedNetEnv("45_intersections", n_envs=B, obs_mode="option3", network=build_network(g, n_replicas=B, rng_seed=3)) for _ in range(2)]
    rng = np.random.default_rng(4)
    for e in envs:
        e.reset()
    for k in range(steps):
        acts = rng.uniform(0, 4, size=(B, envs[0].n_actions))
        obs_h, rew_h, term_h, _, _ = envs[0].step(acts)
        obs_d, rew_d, term_d = envs[1].step_device(torch.as_tensor(acts, device="cuda"))
        assert obs_d.is_cuda and obs_d.dtype == torch.float32 and tuple(obs_d.shape) == obs_h.shape
        assert np.array_equal(obs_d.cpu().numpy(), obs_h) and np.array_equal(rew_d.cpu().numpy(), rew_h) and term_d == term_h
    assert obs_d.data_ptr() == envs[1].network.engine().rl_device_ptr(1)       # a view, not a copy
    with pytest.raises(ValueError):
        envs[1].step_device(torch.zeros((B, envs[1].n_actions), device="cuda", dtype=torch.float32))
    for e in envs:
        e.close()


@pytest.mark.parametrize("randomized,B", [(False, 256), (True, 256), (False, 384), (True, 320)])   # 384 / 320 envs: chains of 256 + 128
def test_forked_rl_chains_give_the_same_episode(randomized, B, monkeypatch):
    """PEDN_RL_CHAINS=2: pedn_rl_step leaves the two halves of the envs stepping as two chains on two streams ACROSS calls and joins
    them only when something needs the whole batch.  Against one chain: device-resident actions stepped asynchronously in stretches,
    an observation fetch (join) after every stretch, a setter and host-side actions (which go through the engine's stream alone) in
    between, a reset and a second episode -- observations, rewards, histories, flags identical."""
    torch = pytest.importorskip("torch")
    g = Golden("rl_i45_opt3")
    steps = 48

    def episode(chains):
        monkeypatch.setenv("PEDN_RL_CHAINS", chains)
        monkeypatch.setenv("PEDN_STREAM_PROBE", "0")
        np.random.seed(5)
        env = VecPedNetEnv("45_intersections", n_envs=B, obs_mode="option3", network=build_network(g, n_replicas=B, rng_seed=3))
        e = env.network.engine()
        gen = torch.Generator(device="cuda").manual_seed(2)
        acts = torch.rand((steps, B, env.n_actions), generator=gen, device="cuda", dtype=torch.float64) * 4.0
        torch.cuda.synchronize()
        row = B * env.n_actions * 8
        out = []
        for ep in range(2):
            env.reset(options={"randomize": True, "mode": "vectorised"} if randomized else None, seed=20 + ep)
            t = 1
            for stretch in (1, 7, 16, 5):
                for _ in range(stretch):                              # asynchronous, nothing fetched: the chains stay forked
                    e.rl_step_device(acts.data_ptr() + (t - 1) * row, t)
                    t += 1
                obs, rew = e.rl_observe(t - 1, accumulate=False)       # joins
                out.append((obs.copy(), rew.copy()))
                if stretch == 7:                                       # a setter between two stretches
                    link = next(iter(env.network.links.values()))
                    link.front_gate_width = link.front_gate_width
                if stretch == 16:                                      # host-side actions: through the engine's own buffer and stream
                    o, r = e.rl_step(acts[t - 1].cpu().numpy(), t, 1)
                    out.append((o.copy(), r.copy()))
                    t += 1
            out.append(tuple(e.read_block(f, 0, t) for f in (0, 2, 5, 9)))
            out.append((e.error_flags()[1],))
        env.close()
        return out

    one, two = episode("1"), episode("2")
    assert len(one) == len(two)
    for a, b in zip(one, two):
        for x, y in zip(a, b):
            assert np.array_equal(x, y)


def test_step_device_without_host_synchronisation_gives_the_same_rollout():
    """step_device(sync=False): the engine's stream waits for the caller's torch stream and vice versa through events only.  A rollout
    whose actions depend on the previous observations (so that any missing dependency would show) equals the synchronous one."""
    torch = pytest.importorskip("torch")
    g = Golden("rl_i45_opt3")
    B, steps = 128, 40

    def rollout(sync):
        env = VecPedNetEnv("45_intersections", n_envs=B, obs_mode="option3", network=build_network(g, n_replicas=B, rng_seed=3))
        env.reset()
        gen = torch.Generator(device="cuda").manual_seed(1)
        obs = torch.zeros((B, env.n_obs), device="cuda")
        total = torch.zeros(B, device="cuda", dtype=torch.float64)
        trace = []
        for _ in range(steps):
            noise = torch.rand((B, env.n_actions), generator=gen, device="cuda", dtype=torch.float64)
            feedback = (obs[:, :env.n_actions].double().abs() % 1.0)           # the policy reads the last observations
            actions = (4.0 * (0.5 * noise + 0.5 * feedback)).contiguous()
            obs, rew, _ = env.step_device(actions, sync=sync)
            total += rew[:, 0].double()
            trace.append(obs.clone())
        torch.cuda.synchronize()
        out = (torch.stack(trace).cpu().numpy(), total.cpu().numpy(), env.network.engine().read_block(0, 0, steps))
        env.close()
        return out

    a, b = rollout(True), rollout(False)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_step_device_aliases_engine_buffers_and_matches_host_step():
    """In this process the engine library was loaded long before torch is imported (the earlier tests created engines):
    engine first, torch second.  Both share the one HIP runtime engine._bind_hip_runtime put in place."""
    pytest.importorskip("torch")
    _step_device_check()


def test_step_device_with_torch_imported_first():
    """The other import order, in a process of its own."""
    import importlib.util
    import os
    import subprocess
    import sys

    if importlib.util.find_spec("torch") is None:
        pytest.skip("torch is not installed")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import torch, sys; torch.zeros(1, device='cuda'); sys.path.insert(0, 'tests'); import test_gpu_rl as t; "
            "t._step_device_check(); print('ok')")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=root, timeout=900)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


@pytest.mark.parametrize("case", ["rl_nine_opt3", "rl_corridor_opt1", "rl_nine_opt5g2"])
def test_recent_history_mode_serves_the_rl_step(case):
    """The batched RL step reads nothing older than the rings of PEDN_HIST_RECENT hold: same observations and rewards."""
    g = Golden(case)
    rl = g.info["rl"]
    net = build_network(g, n_replicas=1, replica_offset=g.replica, rng_seed=g.seed, history="recent")
    env = VecPedNetEnv(g.info["scenario"], n_envs=1, obs_mode=rl["obs_mode"], normalize_obs=rl["normalize"],
                       action_gap=rl["action_gap"], network=net)
    acts, ref_obs, ref_rew = g.state("rl_actions"), g.state("rl_obs"), g.state("rl_rewards")
    for k in range(rl["env_steps"]):
        obs, rew, *_ = env.step(acts[k:k + 1])
        assert np.array_equal(obs[0], ref_obs[k]) and np.array_equal(rew[0], ref_rew[k]), k
    e = net.engine()
    last = g.steps - 1
    ci = e.read_block(LINK_FIELDS["cumulative_inflow"][0], 0, last + 1)[:, :e.n_links, 0].T
    assert np.array_equal(ci, g.state("cumulative_inflow")[:, :last + 1])
    env.reset()
    obs, rew, *_ = env.step(acts[0:1])          # the rings start over with the episode
    assert np.array_equal(obs[0], ref_obs[0])
    env.close()


def _feedback_policy(torch, B, n_actions):
    """A policy whose actions depend on the observations it is handed (any missing dependency between the captured launches would
    show) and on a state of its own that it advances in place (a linear congruential sequence per action slot: replayed or eager, the
    same tensor ops in the same order)."""
    state = (torch.arange(B * n_actions, device="cuda", dtype=torch.int64).reshape(B, n_actions) * 7919 + 17) % 2147483648

    def policy(obs):
        state.mul_(1103515245).add_(12345).remainder_(2147483648)
        noise = state.double() / 2147483648.0
        feedback = obs[:, :n_actions].double().abs() % 1.0
        return (4.0 * (0.5 * noise + 0.5 * feedback)).contiguous()
    return policy


@pytest.mark.parametrize("scenario,history,gap,randomized,spr", [("45_intersections", "full", 1, False, 1), ("45_intersections", "recent", 2, False, 1),
                                                                 ("45_intersections", "full", 1, True, 4), ("nine_intersections", "full", 1, False, 1),
                                                                 ("nine_intersections", "recent", 1, False, 16)])
def test_graph_replayed_rollout_equals_the_eager_one(scenario, history, gap, randomized, spr):
    """the env step with a device-resident step index (pedn_rl_clock_begin / pedn_rl_step_clocked) captured together
    with the policy as ONE torch.cuda.CUDAGraph (spr policy steps per graph) and replayed -- against the same rollout through
    step_device, bit for bit: the observations at the end of each episode, the accumulated rewards, histories, flags; two episodes with a
    lazy reset in between (the second one shorter, so that rows of the first survive above it), an engine read in the middle of an episode
    (ends the clocked section, which the rollout must notice and begin again), the last step of the horizon."""
    torch = pytest.importorskip("torch")
    g = Golden("rl_i45_opt3" if scenario == "45_intersections" else "rl_nine_opt3")
    B = 128

    def rollout(graphed):
        np.random.seed(5)
        env = VecPedNetEnv(scenario, n_envs=B, obs_mode="option3", action_gap=gap, reward_mode="all",
                           network=build_network(g, n_replicas=B, rng_seed=3, history=history))
        e = env.network.engine()
        policy = _feedback_policy(torch, B, env.n_actions)
        total = torch.zeros((B, len(env.possible_agents)), device="cuda", dtype=torch.float64)

        def on_step(obs, rew):
            total.add_(rew.double())

        roll = env.capture(policy, on_step, steps_per_replay=spr) if graphed else None
        T = env.simulation_steps
        out = []
        for ep, n_steps in enumerate((T // gap, 65)):          # 65 = the eager first step + 64: a whole number of graphs of 1, 4 or 16 steps
            env.reset(options={"randomize": True, "mode": "vectorised"} if randomized else None, seed=30 + ep)
            total.zero_()
            looked, done = False, False
            while env.sim_step <= n_steps * gap:
                assert not done
                if graphed:
                    done = roll.step()
                else:
                    obs, rew, done = env.step_device(policy(env.device_views()[0]), sync=False)
                    on_step(obs, rew)
                if env.sim_step > 100 and ep == 0 and not looked:          # something else looks at the engine in mid-episode
                    out.append(torch.as_tensor(e.read_block(2, 50, 52)))
                    looked = True
            assert done == (ep == 0)
            torch.cuda.synchronize()
            out.append(env.device_views()[0].clone())
            out.append(total.clone())
            t_end = env.sim_step - 1
            lo = max(0, t_end - 2)
            out.extend(torch.as_tensor(e.read_block(f, lo if e.history_rows(f) < T + 1 else 0, t_end + 1)) for f in (0, 2, 3, 9, 10))
            out.append(torch.as_tensor(e.error_flags()[1].astype(np.int64)))
        if graphed:
            assert roll.replays >= (T // gap + 64) // spr - 6 and roll.eager_steps <= 3 + 2 * spr, (roll.replays, roll.eager_steps)
            with pytest.raises(IndexError):
                for _ in range(T):
                    roll.step()
        env.close()
        return [x.cpu().numpy() for x in out]

    a, b = rollout(False), rollout(True)
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert np.array_equal(x, y), i


def test_graph_is_captured_again_when_a_randomised_reset_switches_the_kernels():
    """The captured launches carry the engine's device view by value.  A rollout captured during a plain episode must not be replayed
    after reset(options={'randomize': True}) gave every env its own parameters (other kernels, other pointers): the rollout sees the
    launch signature change, captures again, and the episode equals the eager one."""
    torch = pytest.importorskip("torch")
    g = Golden("rl_i45_opt3")
    B, steps = 128, 40

    def run(graphed):
        np.random.seed(5)
        env = VecPedNetEnv("45_intersections", n_envs=B, obs_mode="option3", network=build_network(g, n_replicas=B, rng_seed=3))
        policy = _feedback_policy(torch, B, env.n_actions)
        roll = env.capture(policy) if graphed else None
        out = []
        for ep, options in enumerate((None, {"randomize": True, "mode": "vectorised"}, None)):
            env.reset(options=options, seed=7)
            for _ in range(steps):
                if graphed:
                    roll.step()
                else:
                    env.step_device(policy(env.device_views()[0]), sync=False)
            torch.cuda.synchronize()
            out.append(env.device_views()[0].clone().cpu().numpy())
            out.append(env.network.engine().read_block(2, 0, steps))
        if graphed:
            assert roll.recaptures == 1, roll.recaptures          # plain -> per-replica scenarios; the third episode keeps them
        env.close()
        return out

    for x, y in zip(run(False), run(True)):
        assert np.array_equal(x, y)


def test_clocked_step_argument_errors_and_horizon_guard():
    """pedn_rl_clock_begin refuses a step whose turning fractions are not prepared (the first step of an episode); a clocked step
    enqueued beyond the horizon does nothing; pedn_rl_clock_end returns the next step."""
    torch = pytest.importorskip("torch")
    g = Golden("rl_nine_opt3")
    B = 64
    env = VecPedNetEnv("nine_intersections", n_envs=B, obs_mode="option3", network=build_network(g, n_replicas=B, rng_seed=3))
    e = env.network.engine()
    env.reset()
    with pytest.raises(RuntimeError, match="not prepared"):
        e.rl_clock_begin(1)
    with pytest.raises(RuntimeError, match="pedn_rl_clock_begin"):
        e.rl_step_clocked(0)
    acts = torch.full((B, env.n_actions), 2.0, device="cuda", dtype=torch.float64)
    env.step_device(acts)
    T = env.simulation_steps
    e.rl_clock_begin(2)
    assert e.rl_clocked()
    for _ in range(5):
        e.rl_step_clocked(acts.data_ptr())
    assert e.rl_clock_end() == 7 and not e.rl_clocked()
    ref = e.read_block(2, 0, 7).copy()
    e.rl_clock_begin(7)
    for _ in range(T + 5):                        # runs into the horizon: the surplus launches are idle
        e.rl_step_clocked(acts.data_ptr())
    assert e.rl_clock_end() == T + 1
    assert np.array_equal(e.read_block(2, 0, 7), ref) and e.error_flags()[0] == 0
    assert e.read_block(2, T, T + 1).sum() > 0
    env.close()


def test_reference_reset_determinism_script_shapes():
    """rl/test_reset_determinism.py:29-83,294-330 restated: 45_intersections, option3 without normalisation, five episodes of
    reset(options={'randomize': True}) + empty action dicts.  What the reference's committed run pins
    (rl/outputs/reset_determinism_test/statistics.json): one agent, gate_24, 20 observation values, 5 episodes -- and 701 observations
    per episode (:59-77); its episodes differ by design (max_difference 86.0), so do these."""
    env = PedNetParallelEnv("45_intersections", normalize_obs=False, obs_mode="option3", render_mode=None, verbose=False)
    assert env.possible_agents == ["gate_24"] and env.observation_space("gate_24").shape == (20,)
    episodes = []
    for ep in range(5):
        obs, infos = env.reset(options={"randomize": True})
        states = [obs["gate_24"].copy()]
        done = False
        while not done:
            obs, rewards, terms, truncs, infos = env.step({})
            states.append(obs["gate_24"].copy())
            done = any(terms.values()) or any(truncs.values())
        episodes.append(np.stack(states))
    assert all(e.shape == (701, 20) and np.isfinite(e).all() for e in episodes)
    diffs = [np.abs(episodes[i] - episodes[j]).max() for i in range(5) for j in range(i)]
    assert min(diffs) > 0.0, diffs          # every pair of episodes differs somewhere
    env.close()


def test_pettingzoo_parallel_api_checks_restated():
    """rl/test_pz_api.py:18-47 runs pettingzoo.test.parallel_api_test(env, num_cycles=100) on nine_intersections; pettingzoo is not in
    this image, so the checks that function makes are restated: dict-of-agents returns keyed by the live agents, the agent set inside
    possible_agents, the space accessors returning the SAME object every call, observations inside their spaces, two resets."""
    env = PedNetParallelEnv(dataset="nine_intersections", normalize_obs=True)
    rng = np.random.default_rng(0)
    assert len(env.possible_agents) > 0
    # what the reference's trainers / rule-based agents read off the env (rl/rl_utils.py:133, rl/agents/rule_based.py:189)
    assert env.obs_mode == "option1" and env.obs_builder.features_per_link == 3 and env.normalize_obs is True
    for _ in range(2):
        obs, infos = env.reset()
        assert isinstance(obs, dict) and isinstance(infos, dict)
        assert set(obs) == set(env.agents) == set(infos)
        live = set(env.agents)
        for _cycle in range(100):
            actions = {a: rng.uniform(env.action_space(a).low, env.action_space(a).high).astype(np.float32) for a in env.agents}
            obs, rew, terminated, truncated, info = env.step(actions)
            for v in (obs, rew, terminated, truncated, info):
                assert isinstance(v, dict) and set(v) == live
            assert set(env.agents) <= set(env.possible_agents) and set(env.agents) == live
            for a in env.agents:
                assert env.observation_space(a) is env.observation_space(a) and env.action_space(a) is env.action_space(a)
                o = obs[a]
                assert o.dtype == np.float32 and o.shape == env.observation_space(a).shape and env.observation_space(a).contains(o)
                assert isinstance(rew[a], float) and terminated[a] is False and truncated[a] is False
    with pytest.raises(ValueError):
        env.observation_space("nobody")
    env.close()
