"""The reference's rl/rl_example.py on the MI355X engine: the PettingZoo-style single env (dict of agents in, dicts out) on
nine_intersections with its predefined gater controllers -- two episodes (the second after reset(options={'randomize': True})), random
actions from the agents' action spaces, the reference's output files at the end.

    python examples/rl_example.py [steps per episode]
"""
import sys
import warnings

from _common import ROOT

from rl import PedNetParallelEnv  # noqa: E402  (reference import path)


def test_environment(max_steps=None):
    dataset = "nine_intersections"
    env = PedNetParallelEnv(dataset, render_mode="animate", data_dir=f"{ROOT}/data")
    print(f"Created PedNet environment with dataset: {dataset}; {len(env.agents)} agents: {env.agents}")
    for episode in range(2):
        if episode > 0:
            env.reset(options={"randomize": True})
        total, steps = 0.0, 0
        for step in range(max_steps or env.simulation_steps):
            actions = {}
            for agent_id in env.agents:
                space = env.action_space(agent_id)
                actions[agent_id] = space.low if space.shape == (1,) else space.sample()
            observations, rewards, terminations, truncations, infos = env.step(actions)
            total += sum(rewards.values())
            steps += 1
            if any(terminations.values()) or any(truncations.values()):
                break
        print(f"Episode {episode + 1}: {steps} steps, summed reward {total:.1f}, terminated {any(terminations.values())}")
    env.save(simulation_dir="rl_example", base_dir=f"{ROOT}/outputs")
    print("Environment test completed successfully!")
    return env


if __name__ == "__main__":
    env = test_environment(int(sys.argv[1]) if len(sys.argv) > 1 else None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        env.render(simulation_dir="../outputs/rl_example", vis_actions=True)      # the reference animates here; not provided
    env.close()
