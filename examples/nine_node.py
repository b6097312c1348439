"""The reference's examples/nine_node.py on the MI355X engine: a 3 x 3 grid built directly with Network(adjacency, params, ...), two
origins, three destinations with OD weights (dynamic logit turning fractions), 599 steps.

    python examples/nine_node.py
"""
import warnings

import numpy as np

from _common import save, summary

from src.LTM.network import Network  # noqa: E402  (reference import path)

GRID = np.zeros((9, 9), dtype=int)
for a, b in ((0, 1), (1, 2), (3, 4), (4, 5), (6, 7), (7, 8), (0, 3), (3, 6), (1, 4), (4, 7), (2, 5), (5, 8)):
    GRID[a, b] = GRID[b, a] = 1


def main():
    np.random.seed(0)                       # the origin demand is drawn from numpy's global stream at construction, like the reference
    params = {"unit_time": 10, "simulation_steps": 600, "assign_flows_type": "classic",
              "default_link": {"length": 100, "width": 1, "free_flow_speed": 1.5, "k_critical": 2, "k_jam": 10},
              "demand": {"origin_0": {"peak_lambda": 15, "base_lambda": 5}, "origin_4": {"peak_lambda": 15, "base_lambda": 5}}}
    od_flows = {(0, 8): 5, (4, 8): 10, (0, 3): 5, (4, 3): 1, (0, 1): 5, (4, 1): 1}
    network_env = Network(GRID, params, origin_nodes=[0, 4], destination_nodes=[3, 8, 1], od_flows=od_flows)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        network_env.visualize()             # the reference plots the topology here; this engine warns and carries on
    for t in range(1, params["simulation_steps"]):
        network_env.network_loading(t)
    print("nine_node:", summary(network_env, params["simulation_steps"] - 1))
    node = network_env.nodes[4]
    print("turning fractions of node 4 at the last step:", np.round(np.asarray(node.turning_fractions), 3).tolist())
    print("saved", save(network_env, "nine_node"))


if __name__ == "__main__":
    main()
