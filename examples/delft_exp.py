"""The reference's examples/delft_exp.py loop (real street network of Delft: 298 nodes, 818 links, 140 OD pairs with logit
route choice) on the MI355X engine -- the two compat lines replace nothing else.  The reference needs 99 s for the 499 steps;
one replica here takes ~11 ms, 1024 replicas ~27 ms.  Run from the repository root on a machine with a GPU:

    python examples/delft_exp.py [n_replicas]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pednstream_amd.compat as compat  # noqa: E402

compat.install()

from handlers.output_handler import OutputHandler  # noqa: E402  (reference import paths)
from src.utils.env_loader import NetworkEnvGenerator  # noqa: E402


def main():
    n_replicas = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    env_generator = NetworkEnvGenerator("data")
    network_env = env_generator.create_network("delft", verbose=False, n_replicas=n_replicas)
    network_env.synchronize()          # creates the device engine now (otherwise the first network_loading does, inside the timing)
    start_time = time.time()
    for t in range(1, env_generator.config["params"]["simulation_steps"]):
        network_env.network_loading(t)
    network_env.synchronize()
    print("Simulation time: {:.3f} s for {} replica(s)".format(time.time() - start_time, n_replicas))
    T = env_generator.config["params"]["simulation_steps"]
    busiest = max(network_env.links.values(), key=lambda l: float(l.cumulative_inflow[T - 1]))
    print(f"busiest link {busiest.link_id}: {float(busiest.cumulative_inflow[T - 1]):.0f} pedestrians entered, "
          f"density at the end {float(busiest.density[T - 1]):.3f} ped/m2")
    output_handler = OutputHandler(base_dir="outputs", simulation_dir="delft_paths")
    output_handler.save_network_state(network_env)
    print("saved", output_handler.simulation_dir)


if __name__ == "__main__":
    main()
