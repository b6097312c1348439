"""The calling sequence of the reference's examples/spike.py on the MI355X engine: a network built directly from an adjacency
matrix and a parameter dict, a CUSTOM demand pattern handed over as a callable (registered under its function name and selected
by `params['demand']['origin_N']['pattern']`), turning fractions imposed on the origin's junction, then the stepping loop.

    python examples/spike.py [replicas]

With more than one replica every replica shares the callable's demand and differs by its RNG key."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pednstream_amd.compat as compat  # noqa: E402

compat.install()

from src.LTM.network import Network  # noqa: E402  (reference import path)


def surge_pattern(origin_id, params):
    """Poisson arrivals around a morning bump, a 40-step surge of 30 pedestrians per step, nothing in the last quarter."""
    cfg = params["demand"][f"origin_{origin_id}"]
    T = params["simulation_steps"]
    t = np.arange(T)
    lam = cfg["base_lambda"] + cfg["peak_lambda"] * np.exp(-(t - T / 4) ** 2 / (2 * (T / 20) ** 2))
    demand = np.random.poisson(lam=lam)
    demand[T // 2:T // 2 + 40] = 30
    demand[3 * T // 4:] = 0
    return demand


def main():
    replicas = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    adj = np.array([[0, 1, 1, 1, 0, 0],
                    [1, 0, 1, 1, 0, 0],
                    [1, 1, 0, 1, 0, 0],
                    [1, 1, 1, 0, 1, 0],
                    [0, 0, 0, 1, 0, 1],
                    [0, 0, 0, 0, 1, 0]])
    params = {"unit_time": 10, "simulation_steps": 600, "assign_flows_type": "classic",
              "default_link": {"length": 100, "width": 2, "free_flow_speed": 1.1, "k_critical": 2, "k_jam": 6, "speed_noise_std": 0,
                               "fd_type": "yperman", "bi_factor": 1, "controller_type": "gate"},
              "demand": {"origin_4": {"pattern": "surge_pattern", "peak_lambda": 20, "base_lambda": 10}}}
    np.random.seed(1)
    network_env = Network(adj, params, origin_nodes=[4], demand_pattern=[surge_pattern], n_replicas=replicas)
    network_env.update_turning_fractions_per_node(node_ids=[4], new_turning_fractions=np.array([[1, 0, 0, 1, 0, 1]]))
    for t in range(1, params["simulation_steps"]):
        network_env.network_loading(t)
    entered = float(network_env.links[(4, 3)].cumulative_inflow[params["simulation_steps"] - 1])
    offered = float(np.sum(network_env.nodes[4].demand))
    peak = float(np.max(np.asarray(network_env.links[(4, 3)].density)))
    print(f"surge demand offered at node 4: {offered:.0f} pedestrians, entered link (4,3): {entered:.0f}, its peak density {peak:.2f} ped/m^2")


if __name__ == "__main__":
    main()
