"""The reference's examples/Melbourne.py on the MI355X engine: the melbourne scenario with the origin demand taken from a table of
pedestrian counts per minute through a custom demand callable (create_network(name, [fn])).  The reference reads the counts of the
sensor nearest to the origin from data/melbourne/melbourne.csv; that table is not part of this repository, so the counts here are
synthetic -- the calling sequence (a closure named like the yaml's custom pattern, counts spread over six 10-second steps per minute
and rounded up) is the reference's.

    python examples/melbourne.py [n_replicas]
"""
import sys
import time

import numpy as np

from _common import save, summary

from src.utils.env_loader import NetworkEnvGenerator  # noqa: E402  (reference import path)


def expand_to_10sec(minute_counts):
    return np.repeat(np.asarray(minute_counts, dtype=np.float64) / 6, 6)


def create_demand_function(counts_per_minute):
    def node_demand_from_data(origin_node, params=None, _table=counts_per_minute):
        return np.ceil(expand_to_10sec(_table[origin_node]))
    return node_demand_from_data


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    minutes = np.arange(90)
    table = {289: np.round(45 + 40 * np.sin(minutes / 9.0) ** 2 + (minutes * 31) % 13)}        # the scenario's origin node
    env_generator = NetworkEnvGenerator("data")
    # (the scenario file names no custom pattern for its origin: the override selects the callable by its __name__, od_manager.py:125-143)
    network_env = env_generator.create_network("melbourne", [create_demand_function(table)], n_replicas=R,
                                               demand_params_overrides={"origin_289": {"pattern": "node_demand_from_data"}})
    steps = env_generator.config["params"]["simulation_steps"]
    t0 = time.perf_counter()
    for t in range(1, steps):
        network_env.network_loading(t)
    network_env.engine().synchronize()
    print(f"Simulation time: {time.perf_counter() - t0:.2f}")
    print(f"melbourne x {R}:", summary(network_env, steps - 1))
    print("saved", save(network_env, "melbourne"))


if __name__ == "__main__":
    main()
