"""The reference's rl/train_ppo_sb3.py with the batch where it has a vector of ONE env (rl/train_ppo_sb3.py:246: DummyVecEnv([env_fn])):
``PedNetSB3VecEnv`` is the reference's single-agent wrapper (:49-141) and SB3's VecEnv in one object for ``n_envs`` replicas on the GPU.

    python examples/train_ppo_sb3.py [dataset] [n_envs] [total_timesteps]

With stable-baselines3 installed it trains PPO for ``total_timesteps``; without it (this build image) it drives the same VecEnv protocol
-- step_async / step_wait with automatic resets -- with a random policy for two episodes and reports what SB3's rollout collection would see.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pednstream_amd.sb3_env import PedNetSB3VecEnv  # noqa: E402


def main():
    dataset = sys.argv[1] if len(sys.argv) > 1 else "45_intersections"
    n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 256
    total_timesteps = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
    env = PedNetSB3VecEnv(dataset, n_envs=n_envs, normalize_obs=True, randomize=True, action_gap=10, seed=0,
                          data_dir=os.path.join(ROOT, "data"), history="recent")
    print(f"Observation space: {env.observation_space.shape}  Action space: {env.action_space.shape}  envs: {env.num_envs}")
    try:
        from stable_baselines3 import PPO
        from stable_baselines3.common.vec_env import VecMonitor
    except ImportError:
        PPO = None
    if PPO is not None:      # the reference's call (:285-305) with the rollout length counted per env
        model = PPO("MlpPolicy", VecMonitor(env), learning_rate=3e-4, n_steps=max(8, 2048 // n_envs), batch_size=64, n_epochs=10,
                    gamma=0.99, gae_lambda=0.95, clip_range=0.2, ent_coef=0.01, vf_coef=0.5, max_grad_norm=0.5, verbose=1)
        model.learn(total_timesteps=total_timesteps)
        env.close()
        return
    rng = np.random.default_rng(0)
    obs = env.reset()
    episodes, ret, returns = 0, np.zeros(n_envs), []
    t0, n = time.perf_counter(), 0
    while episodes < 2:
        actions = rng.uniform(env.action_space.low, env.action_space.high, size=(n_envs,) + env.action_space.shape).astype(np.float32)
        env.step_async(actions)
        obs, rewards, dones, infos = env.step_wait()
        ret += rewards
        n += 1
        if dones.all():
            assert "terminal_observation" in infos[0] and obs.shape == (n_envs,) + env.observation_space.shape
            returns.append(float(ret.mean()))
            ret[:] = 0.0
            episodes += 1
    dt = time.perf_counter() - t0
    print(f"stable-baselines3 is not installed: random policy through the VecEnv protocol, {episodes} episodes of {n // episodes} policy "
          f"steps x {n_envs} envs, mean episode returns {returns}, {n_envs * n / dt:.3g} env-steps/s (host policy, automatic randomised resets)")
    env.close()


if __name__ == "__main__":
    main()
