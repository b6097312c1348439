"""Per-launch durations (dispatch timestamps, pedn_profile_step) of a network x 1024 replicas under the bench's demand, for
A/B runs inside ONE gpurun call: boxes differ by +-1 us per kernel, so builds and launch plans are compared on one box.

    PEDN_FUSE_TP=0 python tools/kernel_times.py melbourne delft
    PEDN_HIP_LIB=$PWD/pednstream_amd/csrc/libpedn_hip_other.so python tools/kernel_times.py melbourne
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import replica_demand  # noqa: E402
from pednstream_amd import NetworkEnvGenerator  # noqa: E402

DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "data")
tag = " ".join(f"{k}={os.environ[k]}" for k in ("PEDN_FUSE_TP", "PEDN_LINK_OWNER", "PEDN_INLINE_TF", "PEDN_NODE_MD", "PEDN_STREAMS") if k in os.environ)
lib = os.path.basename(os.environ.get("PEDN_HIP_LIB", "libpedn_hip.so"))
for network in sys.argv[1:]:
    R = 1024
    net = NetworkEnvGenerator(DATA).create_network(network, verbose=False, n_replicas=R, rng_seed=0)
    e = net.engine()
    T = net.simulation_steps
    for nid in net.origin_nodes:
        net.set_demand_matrix(nid, np.stack([replica_demand(T, r) for r in range(R)]))
    e.run(1, 200)
    e.synchronize()
    prof = np.array([e.profile_step(t) for t in range(200, 260)])
    print(f"{network:10s} {lib} {tag or 'default plan':24s} node_kernel {prof[:, 1].mean() * 1e3:6.2f} us   second launch "
          f"(link update and / or turning fractions) {prof[:, 2].mean() * 1e3:6.2f} us   sum {(prof[:, 1] + prof[:, 2]).mean() * 1e3:6.2f} us")
    net.close()
