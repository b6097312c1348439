"""GPU tests of the SB3-shaped batched env (pednstream_amd/sb3_env.py): the reference's single-agent wrapper around its dict-of-agents
env (rl/train_ppo_sb3.py:49-141: observations / actions concatenated in agent order, reward = sum over the agents) + SB3's VecEnv
protocol (automatic reset with terminal_observation), for n_envs replicas at once."""
import numpy as np
import pytest

from pednstream_amd.rl_env import PedNetParallelEnv, VecPedNetEnv
from pednstream_amd.sb3_env import PedNetSB3VecEnv

pytestmark = pytest.mark.gpu


def _wrapper_step(env, agents, action):
    """The reference's PedNetSB3Wrapper.step (rl/train_ppo_sb3.py:101-132) restated on the dict-of-agents env."""
    acts, k = {}, 0
    for a in agents:
        n = env.action_space(a).shape[0]
        acts[a] = action[k:k + n]
        k += n
    obs, rewards, terms, truncs, infos = env.step(acts)
    return (np.concatenate([obs[a] for a in agents], dtype=np.float32), sum(rewards.values()), any(terms.values()),
            any(truncs.values()), dict(infos[agents[0]], individual_rewards=rewards))


def test_one_env_equals_the_reference_wrapper_around_the_single_env():
    kw = dict(normalize_obs=True, obs_mode="option3", action_gap=10, seed=3)
    np.random.seed(5)          # the scenario's demand is drawn from numpy's global stream at construction (od_manager.py:101-155)
    sb3 = PedNetSB3VecEnv("45_intersections", n_envs=1, detailed_infos=True, **kw)
    np.random.seed(5)
    one = PedNetParallelEnv("45_intersections", **kw)
    agents = one.possible_agents
    assert sb3.num_envs == 1 and sb3.observation_space.shape == (sum(one.observation_space(a).shape[0] for a in agents),)
    assert np.array_equal(sb3.action_space.low, np.concatenate([one.action_space(a).low for a in agents]))
    assert np.array_equal(sb3.action_space.high, np.concatenate([one.action_space(a).high for a in agents]))
    o1 = sb3.reset()
    o2, _ = one.reset()
    assert o1.dtype == np.float32 and np.array_equal(o1[0], np.concatenate([o2[a] for a in agents]))
    rng = np.random.default_rng(0)
    n_steps = one.simulation_steps // 10
    for k in range(n_steps):
        a = rng.uniform(sb3.action_space.low, sb3.action_space.high).astype(np.float32)
        obs, rew, dones, infos = sb3.step(a[None])
        wo, wr, wt, wtr, winfo = _wrapper_step(one, agents, a)
        assert rew.dtype == np.float32 and rew[0] == np.float32(wr), (k, rew, wr)
        assert dones.dtype == bool and bool(dones[0]) == (wt or wtr)
        assert infos[0]["individual_rewards"] == winfo["individual_rewards"]
        assert infos[0]["cumulative_reward"] == winfo["cumulative_reward"] and infos[0]["step"] == winfo["step"]
        if not dones[0]:
            assert np.array_equal(obs[0], wo), k
        else:   # DummyVecEnv's contract: the last observation in the info, the first one of the next episode returned
            assert k == n_steps - 1
            assert np.array_equal(infos[0]["terminal_observation"], wo) and infos[0]["TimeLimit.truncated"] is False
            r2, _ = one.reset()
            assert np.array_equal(obs[0], np.concatenate([r2[a] for a in agents]))
    sb3.close()
    one.close()


def test_batch_of_envs_protocol_and_automatic_reset():
    B = 6
    kw = dict(normalize_obs=False, obs_mode="option1", action_gap=25, seed=11, reward_mode="all")
    np.random.seed(6)
    sb3 = PedNetSB3VecEnv("nine_intersections", n_envs=B, **kw)
    np.random.seed(6)
    twin = VecPedNetEnv("nine_intersections", n_envs=B, **kw)
    with pytest.raises(RuntimeError):
        sb3.step_wait()
    with pytest.raises(ValueError):
        sb3.step_async(np.zeros((B, sb3.action_space.shape[0] + 1)))
    assert sb3.get_attr("render_mode") == [None] * B and sb3.env_is_wrapped(object) == [False] * B
    assert sb3.get_attr("simulation_steps", indices=[0, 2]) == [twin.simulation_steps] * 2
    assert sb3.seed(5) == [5 + i for i in range(B)]
    obs = sb3.reset()
    tobs, _ = twin.reset()
    assert obs.shape == (B, twin.n_obs) and np.array_equal(obs, tobs)
    rng = np.random.default_rng(1)
    n_steps = twin.simulation_steps // 25
    for k in range(n_steps + 3):        # ... and into the next episode
        a = rng.uniform(twin.action_low, twin.action_high, (B, twin.n_actions)).astype(np.float32)
        sb3.step_async(a)
        obs, rew, dones, infos = sb3.step_wait()
        if twin.sim_step + twin.action_gap - 1 > twin.simulation_steps:
            twin.reset()
        to, tr, term, _, _ = twin.step(a.astype(np.float64))
        assert np.array_equal(rew, tr.astype(np.float64).sum(axis=1).astype(np.float32)) and rew.shape == (B,)
        assert dones.shape == (B,) and dones.all() == term and dones.any() == term and len(infos) == B
        if term:
            assert all(np.array_equal(infos[i]["terminal_observation"], to[i]) for i in range(B))
            first, _ = twin.reset()
            assert np.array_equal(obs, first)
        else:
            assert np.array_equal(obs, to) and infos == [{} for _ in range(B)]
    sb3.close()
    twin.close()


def test_randomised_resets_draw_new_scenarios():
    sb3 = PedNetSB3VecEnv("45_intersections", n_envs=4, randomize=True, action_gap=100, seed=2, history="recent")
    sb3.seed(7)
    sb3.reset()
    first = sb3.vec.scenarios
    assert first is not None                       # reset(options={'randomize': True}) went through VecPedNetEnv.randomize
    kc_a = sb3.vec.network.engine().get_link_params()["kc"].copy()
    a = np.tile((sb3.vec.action_low + sb3.vec.action_high)[None] / 2, (4, 1))
    for _ in range(sb3.vec.simulation_steps // 100):
        obs, rew, dones, infos = sb3.step(a)
    assert dones.all() and "terminal_observation" in infos[0]
    assert sb3.vec.sim_step == 1 and sb3.vec.scenarios is not None
    kc_b = sb3.vec.network.engine().get_link_params()["kc"]
    assert not np.array_equal(kc_a, kc_b)          # a new draw per episode
    sb3.close()
