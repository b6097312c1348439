#!/usr/bin/env python3
"""Measured packing cost of every node of a scenario -> data/<scenario>/pack_cost.json (read by NetworkEnvGenerator.create_network,
handed to pedn_create as pedn_model_desc.node_cost: nodes of similar cost then share a node-kernel workgroup).

Needs the profiling build (`make -C pednstream_amd/csrc phase-profile`) and a GPU:

    python tools/pack_calibrate.py delft melbourne 45_intersections

The cost of a node = mean ticks (s_memtime) its slowest slot wave needs from kernel entry to the node kernel's first block barrier, over
200 steps of 1024 replicas under the bench's per-replica demand (see tools/pack_analysis.py for what that explains)."""
import ctypes
import json
import os
import sys

from pack_analysis import LIB, ROOT, measure, node_costs


def main():
    lib = ctypes.CDLL(LIB)
    for network in sys.argv[1:] or ["delft"]:
        net, arrive, wait, life = measure(lib, network)
        bins, deg, slot_cost, cost, _ = node_costs(net.engine().model, arrive)
        ids = net.engine().model["node_id"]
        out = {"node_cost": {str(int(ids[n])): round(c, 1) for n, c in sorted(cost.items())},
               "unit": "s_memtime ticks from kernel entry to node_kernel's first barrier, slowest slot wave of the node, mean of 200 steps",
               "calibration": "tools/pack_calibrate.py: 1024 replicas, steps 150..349, bench.py's per-replica Poisson demand, static packing, one chain"}
        path = os.path.join(ROOT, "data", network, "pack_cost.json")
        with open(path, "w") as f:
            json.dump(out, f)
        print(f"{network}: {len(cost)} nodes, cost {min(cost.values()):.0f} .. {max(cost.values()):.0f} ticks -> {os.path.relpath(path, ROOT)}")
        net.close()


if __name__ == "__main__":
    main()
