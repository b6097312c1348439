"""The reference's examples/six_node.py loop (build od_flow_example, step, close a gate for nine steps) on the MI355X
engine, unchanged apart from the two compat lines.  Run from the repository root on a machine with a GPU:

    python examples/six_node.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pednstream_amd.compat as compat  # noqa: E402

compat.install()

from src.utils.env_loader import NetworkEnvGenerator  # noqa: E402  (reference import path)


def main():
    env_generator = NetworkEnvGenerator("data")
    network_env = env_generator.create_network("od_flow_example")
    for t in range(1, env_generator.config["params"]["simulation_steps"]):
        network_env.network_loading(t)
        if t in [100, 101, 102, 103, 104, 105, 106, 107, 108]:
            network_env.links[(3, 5)].back_gate_width -= 0.1
    total = sum(float(l.cumulative_inflow[499]) for l in network_env.links.values())
    print(f"sum of cumulative inflow at t=499: {total:.0f}")
    print(f"link (3,5): cumulative_inflow[499] = {network_env.links[(3, 5)].cumulative_inflow[499]:.0f}, "
          f"density[499] = {network_env.links[(3, 5)].density[499]:.3f}, back gate = {network_env.links[(3, 5)].back_gate_width:.3f}")

    from pednstream_amd.output_handler import OutputHandler

    out = OutputHandler(base_dir="outputs", simulation_dir="six_node_exp")
    out.save_network_state(network_env)
    print("saved", out.simulation_dir)


if __name__ == "__main__":
    main()
