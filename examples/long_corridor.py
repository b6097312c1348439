"""The reference's examples/long_corridor.py on the MI355X engine: six nodes in a row, pedestrians entering from both ends (counter
flow on every corridor), 599 steps -- then its commented-out second scenario as a variant: a bottleneck link in the middle, imposed
turning fractions, demand switched off in place (`node.demand[...] = 0` writes through to the device before the next step).

    python examples/long_corridor.py
"""
import numpy as np

from _common import save, summary

from src.LTM.network import Network  # noqa: E402  (reference import path)

LINE = np.zeros((6, 6), dtype=int)
for a in range(5):
    LINE[a, a + 1] = LINE[a + 1, a] = 1


def counter_flow():
    np.random.seed(0)
    params = {"unit_time": 10, "simulation_steps": 600,
              "default_link": {"length": 100, "width": 2, "free_flow_speed": 1.1, "k_critical": 2, "k_jam": 6, "fd_type": "yperman",
                               "bi_factor": 1, "controller_type": "gate"},
              "demand": {"origin_0": {"peak_lambda": 25, "base_lambda": 5}, "origin_5": {"peak_lambda": 25, "base_lambda": 5}}}
    network_env = Network(LINE, params, origin_nodes=[5, 0])
    for t in range(1, params["simulation_steps"]):
        network_env.network_loading(t)
    print("long_corridor, counter flow:", summary(network_env, params["simulation_steps"] - 1))
    print("saved", save(network_env, "long_corridor"))


def bottleneck():
    np.random.seed(1)
    link = {"length": 50, "width": 2, "free_flow_speed": 1.1, "k_critical": 1, "k_jam": 6, "activity_probability": 0, "fd_type": "yperman",
            "bi_factor": 1, "speed_noise_std": 0, "controller_type": "gate"}
    params = {"unit_time": 10, "simulation_steps": 1200, "assign_flows_type": "classic", "default_link": link,
              "links": {"2_3": dict(link, width=1, k_critical=2)},
              "demand": {"origin_3": {"peak_lambda": 20, "base_lambda": 8}, "origin_2": {"peak_lambda": 20, "base_lambda": 8}}}
    network_env = Network(LINE, params, origin_nodes=[2, 3])
    network_env.update_turning_fractions_per_node(node_ids=[2, 3], new_turning_fractions=np.array([[0, 1, 0.5, 0.5, 0, 1],
                                                                                                    [1, 0, 0, 1, 0.5, 0.5]]))
    network_env.nodes[2].demand[0:10] = 0           # in-place edits of node.demand reach the device before the next step
    network_env.nodes[3].demand[40:] = 0
    for t in range(1, params["simulation_steps"]):
        network_env.network_loading(t)
        if t == 120:
            network_env.links[(3, 4)].back_gate_width = 1
    print("long_corridor, bottleneck 2-3:", summary(network_env, params["simulation_steps"] - 1))


if __name__ == "__main__":
    counter_flow()
    bottleneck()
