"""The calling sequence of the reference's examples/forky_queues.py on the MI355X engine: a five-node fork built directly from
an adjacency matrix, turning fractions imposed on the junction, the FRONT gate of the bottleneck link narrowed before the first
step and opened again during the run (the setter writes through to HBM and mirrors to the reverse link's back gate).

    python examples/forky_queues.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pednstream_amd.compat as compat  # noqa: E402

compat.install()

from src.LTM.network import Network  # noqa: E402  (reference import path)


def main():
    adj = np.array([[0, 1, 0, 0, 0],
                    [1, 0, 1, 0, 1],
                    [0, 1, 0, 1, 0],
                    [0, 0, 1, 0, 0],
                    [0, 1, 0, 0, 0]])
    link = {"length": 100, "width": 3, "free_flow_speed": 1.5, "k_critical": 2, "k_jam": 6, "gamma": 0, "speed_noise_std": 0.05,
            "fd_type": "yperman", "bi_factor": 1.2}
    params = {"unit_time": 10, "simulation_steps": 700, "assign_flows_type": "classic", "default_link": link,
              "links": {"1_2": dict(link, width=1, controller_type="gate"), "2_3": dict(link, length=50, width=1)},
              "demand": {"origin_0": {"peak_lambda": 15, "base_lambda": 5}, "origin_4": {"peak_lambda": 15, "base_lambda": 5}}}
    np.random.seed(2)
    net = Network(adj, params, origin_nodes=[0, 4])
    net.update_turning_fractions_per_node(node_ids=[1], new_turning_fractions=np.array([[1, 0, 0.5, 0.5, 0, 1]]))
    net.links[(1, 2)].front_gate_width = 0.5
    assert net.links[(2, 1)].back_gate_width == 0.5
    queue = []
    for t in range(1, params["simulation_steps"]):
        net.network_loading(t)
        if t == 400:
            net.links[(1, 2)].front_gate_width = 3
        if t in (399, 699):
            queue.append(float(net.links[(1, 2)].num_pedestrians[t]))
    print(f"pedestrians on the bottleneck link (1,2): {queue[0]:.0f} at t=399 behind the half-metre gate, {queue[1]:.0f} at t=699 after it was opened; "
          f"left through (2,3): {net.links[(2, 3)].cumulative_outflow[699]:.0f}")


if __name__ == "__main__":
    main()
