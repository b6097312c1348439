"""The batched env in the shape Stable-Baselines3 trains on (BASELINE config #5: "2048 parallel envs feeding PPO rollout").

The reference feeds SB3 ONE env: ``PedNetSB3Wrapper`` (rl/train_ppo_sb3.py:49-141) turns the dict-of-agents env into a single-agent one
-- observations and actions of all agents concatenated in agent order, reward = sum over the agents, done = any agent done -- and
``DummyVecEnv([env_fn])`` (:246) makes a vector of length one out of it.  ``PedNetSB3VecEnv`` is that pair for ``n_envs`` replicas at
once: the same concatenation (it IS the row layout of ``VecPedNetEnv``), the same sum, and SB3's ``VecEnv`` protocol
(stable_baselines3/common/vec_env/base_vec_env.py: ``reset() -> obs``, ``step_async`` / ``step_wait() -> (obs, rewards, dones, infos)``,
automatic reset of finished envs with ``infos[i]["terminal_observation"]``, ``get_attr`` / ``set_attr`` / ``env_method`` /
``env_is_wrapped`` / ``seed`` / ``set_options``) on top of one ``pedn_rl_step`` per call:

    env = PedNetSB3VecEnv("45_intersections", n_envs=2048, randomize=True)
    model = PPO("MlpPolicy", VecMonitor(env), n_steps=128, ...)          # instead of DummyVecEnv([make_env(...)])

It subclasses ``stable_baselines3.common.vec_env.VecEnv`` when that package is installed and is a plain class with the same methods
when it is not (neither SB3 nor gymnasium is needed to step it).  Every env of the batch shares the horizon, so all of them finish --
and are reset -- in the same call.
"""
import numpy as np

from .rl_env import VecPedNetEnv, _make_box

try:  # pragma: no cover - not installed in the build image
    from stable_baselines3.common.vec_env import VecEnv as _Base
except Exception:  # noqa: BLE001
    _Base = object


class PedNetSB3VecEnv(_Base):
    def __init__(self, dataset, n_envs=1, randomize=False, normalize_obs=True, obs_mode="option1", action_gap=1, seed=0,
                 detailed_infos=False, **kw):
        """``randomize``: every reset draws a new scenario per env (``reset(options={'randomize': True})``: the reference's random
        stream up to 64 envs, drawn on the device above -- ``VecPedNetEnv.randomize``); the reference's ``make_env(randomize=...)``
        accepts the flag and never uses it (rl/train_ppo_sb3.py:143-169).  ``detailed_infos``: every step's info dicts carry what the
        reference's wrapper reports (``step``, ``cumulative_reward`` of the first agent, ``individual_rewards``) -- 2 us per env per
        step of host time; off, a step's infos are empty dicts except at the end of an episode.  Other keywords go to ``VecPedNetEnv``
        (``reward_mode``, ``history``, ``data_dir``, ``device`` ...)."""
        self.vec = VecPedNetEnv(dataset, n_envs=n_envs, obs_mode=obs_mode, normalize_obs=normalize_obs, action_gap=action_gap,
                                seed=seed, **kw)
        self.agents = list(self.vec.possible_agents)
        self.randomize, self.detailed_infos = bool(randomize), bool(detailed_infos)
        observation_space = _make_box(-np.inf, np.inf, (self.vec.n_obs,))                     # :62-67
        action_space = _make_box(self.vec.action_low, self.vec.action_high, (self.vec.n_actions,))   # :70-79
        self.render_mode = None
        if _Base is object:
            self.num_envs, self.observation_space, self.action_space = int(n_envs), observation_space, action_space
            self.reset_infos = [{} for _ in range(self.num_envs)]
            self._seeds = [None] * self.num_envs
            self._options = [{} for _ in range(self.num_envs)]
        else:  # pragma: no cover
            super().__init__(int(n_envs), observation_space, action_space)
        self._actions = None
        self._cumulative = np.zeros(self.num_envs, dtype=np.float64)    # first agent's cumulative reward (the wrapper's combined_info, :128)
        self._episode_seed = None

    # ------------------------------------------------------------------------------------------------ VecEnv protocol
    def seed(self, seed=None):
        """SB3: seeds for the next reset.  One batch, one generator: the first seed keys the next randomised draw."""
        self._episode_seed = seed
        self._seeds = [None if seed is None else seed + i for i in range(self.num_envs)]
        return list(self._seeds)

    def set_options(self, options=None):
        """SB3: options for the next reset -- a dict for all envs or a list (the first entry counts: one batch)."""
        if options is None:
            options = {}
        self._options = list(options) if isinstance(options, (list, tuple)) else [dict(options) for _ in range(self.num_envs)]

    def _reset_batch(self):
        opts = dict(self._options[0]) if self._options and self._options[0] else {}
        if self.randomize:
            opts.setdefault("randomize", True)
        obs, _ = self.vec.reset(options=opts or None, seed=self._episode_seed)
        self._episode_seed = None
        self._seeds = [None] * self.num_envs
        self._options = [{} for _ in range(self.num_envs)]
        self._cumulative[:] = 0.0
        return np.array(obs, dtype=np.float32)

    def reset(self):
        obs = self._reset_batch()
        self.reset_infos = [self._info(i, None) if self.detailed_infos else {} for i in range(self.num_envs)]
        return obs

    def step_async(self, actions):
        a = np.asarray(actions, dtype=np.float64)
        if a.shape != (self.num_envs, self.vec.n_actions):
            raise ValueError(f"actions must have shape {(self.num_envs, self.vec.n_actions)}, got {a.shape}")
        self._actions = a

    def step_wait(self):
        if self._actions is None:
            raise RuntimeError("step_wait() without step_async()")
        a, self._actions = self._actions, None
        obs, rew, terminated, truncated, _ = self.vec.step(a)
        obs = np.array(obs, dtype=np.float32)
        total = np.zeros(self.num_envs, dtype=np.float64)          # sum(rewards.values()), agent after agent (:121)
        for k in range(rew.shape[1]):
            total += rew[:, k]
        self._cumulative += rew[:, 0]
        done = bool(terminated or truncated)
        dones = np.full(self.num_envs, done, dtype=bool)
        infos = [self._info(i, rew) for i in range(self.num_envs)] if self.detailed_infos else [{} for _ in range(self.num_envs)]
        if done:                                                   # DummyVecEnv.step_wait: keep the last observation, start the next episode
            for i, info in enumerate(infos):
                info["TimeLimit.truncated"] = bool(truncated and not terminated)
                info["terminal_observation"] = obs[i]
            obs = self._reset_batch()
            self.reset_infos = [{} for _ in range(self.num_envs)]
        return obs, total.astype(np.float32), dones, infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def _info(self, i, rew):
        """The wrapper's combined_info (:128-130): the first agent's info + every agent's reward."""
        info = {"step": self.vec.sim_step, "cumulative_reward": float(self._cumulative[i])}
        if rew is not None:
            info["individual_rewards"] = {a: float(rew[i, k]) for k, a in enumerate(self.agents)}
        return info

    def close(self):
        self.vec.close()

    def _indices(self, indices):
        if indices is None:
            return list(range(self.num_envs))
        return [indices] if isinstance(indices, int) else list(indices)

    def get_attr(self, attr_name, indices=None):
        """The batch is ONE object: every index sees the adapter's (or the batched env's) attribute."""
        src = self if hasattr(type(self), attr_name) or attr_name in self.__dict__ else self.vec
        value = getattr(src, attr_name)
        return [value for _ in self._indices(indices)]

    def set_attr(self, attr_name, value, indices=None):
        setattr(self, attr_name, value)

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        out = getattr(self.vec, method_name)(*method_args, **method_kwargs)
        return [out for _ in self._indices(indices)]

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._indices(indices)]

    def get_images(self):
        return [None for _ in range(self.num_envs)]

    def render(self, mode=None):
        """Plotting is outside the hot path (``PedNetParallelEnv.render``)."""
        return None
