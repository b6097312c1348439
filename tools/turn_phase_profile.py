"""Where the lifetime of a turn_frac wave goes (profiling build, `make -C pednstream_amd/csrc phase-profile`):
s_memtime stamps per row of a dynamic node, replica group 0, averaged over 100 steps.

    PEDN_FUSE_TP=0 python tools/turn_phase_profile.py delft
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "pednstream_amd", "csrc", "libpedn_hip_phase.so")
os.environ["PEDN_HIP_LIB"] = LIB

from bench import replica_demand  # noqa: E402
from pednstream_amd import NetworkEnvGenerator  # noqa: E402


def main():
    lib = ctypes.CDLL(LIB)
    network = sys.argv[1] if len(sys.argv) > 1 else "delft"
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    net = NetworkEnvGenerator(os.path.join(ROOT, "data")).create_network(network, verbose=False, n_replicas=R, rng_seed=0)
    e = net.engine()
    for nid in net.origin_nodes:
        e.set_demand_matrix(net.nodes[nid].index, np.stack([replica_demand(net.simulation_steps, r) for r in range(R)]))
    e.run(1, 150)
    e.synchronize()
    lib.pedn_debug_tphases(None, 1)
    e.run(150, 250)
    e.synchronize()
    out = (ctypes.c_ulonglong * (4096 * 8))()
    lib.pedn_debug_tphases(out, 0)
    o = np.array(out[:], dtype=np.float64).reshape(4096, 8)
    o = o[o[:, 4] > 0]
    n = o[:, 4:5]
    ph = o[:, :3] / n
    print(f"== {network} x {R}: {len(o)} rows; ticks of s_memtime (shader clock, about 2.4 GHz: 1000 ticks = 0.42 us) per wave, mean over {int(n[0, 0])} launches")
    print("   row  groups products | records+phase1  phase2  phase3 | total")
    order = np.argsort(-ph.sum(axis=1))
    for i in list(order[:12]) + list(order[-4:]):
        print(f"   {i:4d} {int(o[i, 5]):6d} {int(o[i, 6]):8d} | {ph[i, 0]:14.0f} {ph[i, 1]:7.0f} {ph[i, 2]:7.0f} | {ph[i].sum():6.0f}")
    print(f"   mean over rows: {ph.mean(axis=0).round(0).tolist()}  max total {ph.sum(axis=1).max():.0f}")
    net.close()


if __name__ == "__main__":
    main()
