"""How do the launches of the two chains overlap?  (pedn_profile_timeline: dispatch timestamps of every launch of a range)

    python tools/chain_timeline.py melbourne 1024 [steps]

Prints, for a few steps in the middle of the range, when each launch of each chain starts and ends (us after the range's first
launch), and over the whole range: the fraction of the time in which 0 / 1 / 2 launches were running, and the mean duration of
node_kernel / the second launch when running alone and when overlapped."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import replica_demand  # noqa: E402
from pednstream_amd import NetworkEnvGenerator  # noqa: E402

KIND = {0: "turn_frac", 1: "node", 2: "second"}


def main():
    network = sys.argv[1] if len(sys.argv) > 1 else "melbourne"
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    net = NetworkEnvGenerator(os.path.join(ROOT, "data")).create_network(network, verbose=False, n_replicas=R, rng_seed=0)
    e = net.engine()
    for nid in net.origin_nodes:
        net.set_demand_matrix(nid, np.stack([replica_demand(net.simulation_steps, r) for r in range(R)]))
    e.set_streams(2)
    e.run(1, 150)
    e.synchronize()
    rows, chains = e.profile_timeline(150, 150 + steps)
    assert chains == 2
    rows[:, 3:] *= 1e3                               # us
    span = rows[:, 4].max()
    print(f"== {network} x {R}, two chains, steps 150..{150 + steps - 1}: {span / steps:.2f} us per step by the launches' own timestamps")
    mid = 150 + steps // 2
    print("   step chain kernel    start      end   duration")
    for r in rows[(rows[:, 0] >= mid) & (rows[:, 0] < mid + 3)]:
        print(f"   {int(r[0]):4d} {int(r[1]):5d} {KIND[int(r[2])]:7s} {r[3]:8.1f} {r[4]:8.1f} {r[4] - r[3]:8.1f}")
    # occupancy of the time line by number of launches in flight (0.1 us grid)
    grid = np.arange(0.0, span, 0.1)
    running = np.zeros(len(grid), dtype=int)
    nodes = np.zeros(len(grid), dtype=int)
    for r in rows:
        m = (grid >= r[3]) & (grid < r[4])
        running[m] += 1
        if int(r[2]) == 1:
            nodes[m] += 1
    for k in range(0, running.max() + 1):
        print(f"   {k} launches in flight: {100 * (running == k).mean():5.1f} % of the time")
    print(f"   two node_kernels at once: {100 * (nodes == 2).mean():5.1f} %, a node_kernel next to a second launch: {100 * ((nodes == 1) & (running == 2)).mean():5.1f} %")
    for kind in (1, 2):
        sel = rows[rows[:, 2] == kind]
        print(f"   {KIND[kind]:7s}: mean duration {np.mean(sel[:, 4] - sel[:, 3]):6.2f} us over {len(sel)} launches")
    net.close()


if __name__ == "__main__":
    main()
