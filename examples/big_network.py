"""The reference's examples/big_network.py and big_network_directions.py on the MI355X engine: the delft street graph (298 nodes, 818
directed links) built directly from its adjacency matrix with uniform 50 m links -- first with two origins and no destinations
(static uniform turning fractions), then with five origins, four destinations and route choice.

    python examples/big_network.py [n_replicas]
"""
import sys
import time

import numpy as np

from _common import ROOT, save, summary

from src.LTM.network import Network  # noqa: E402  (reference import paths)
from src.utils.env_loader import NetworkEnvGenerator  # noqa: E402


def delft_graph():
    """Adjacency matrix and node positions of data/delft (the reference reads adj_matrix.npy / node_positions.json; this repository
    keeps the same numbers in the scenario directory's topology file)."""
    data = NetworkEnvGenerator(f"{ROOT}/data").load_network_data("delft")
    return np.asarray(data["adjacency_matrix"]), data.get("node_positions")


def run(network_env, steps, label):
    t0 = time.perf_counter()
    for t in range(1, steps):
        network_env.network_loading(t)
    network_env.engine().synchronize()
    print(f"{label}: {steps - 1} x network_loading in {time.perf_counter() - t0:.3f} s;", summary(network_env, steps - 1))


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    adj, pos = delft_graph()
    link = {"length": 50, "width": 1, "free_flow_speed": 1.5, "k_critical": 2, "k_jam": 10}
    np.random.seed(0)
    plain = Network(adj, {"unit_time": 10, "simulation_steps": 500, "default_link": link}, origin_nodes=[0, 8], pos=pos, n_replicas=R)
    run(plain, 500, f"big_network x {R}")
    print("saved", save(plain, "delft"))
    plain.close()
    np.random.seed(0)
    params = {"unit_time": 10, "simulation_steps": 500, "assign_flows_type": "classic", "default_link": dict(link, activity_probability=0.0),
              "demand": {"origin_136": {"peak_lambda": 25, "base_lambda": 5}}}
    routed = Network(adj, params, origin_nodes=[136, 0, 5, 177, 29], destination_nodes=[8, 100, 213, 69], pos=pos, n_replicas=R)
    run(routed, 500, f"big_network_directions x {R}")
    print(f"{len(routed.path_finder.od_paths)} OD pairs with paths; saved", save(routed, "delft_directions"))
    routed.close()


if __name__ == "__main__":
    main()
