"""Two chains of launches against one (pedn_set_streams): random scenarios from tests/fuzz_cases.py with 256 replicas, run
twice on one engine -- the whole batch on one stream, then its two halves on two streams, the run cut into several calls --
every field of every replica, the turning fractions and the error flags bit for bit.

    python tools/gpu_fuzz_chains.py 12000 12100
"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from fuzz_cases import random_case  # noqa: E402
from golden_util import ALL_FIELDS  # noqa: E402
from pednstream_amd import Network  # noqa: E402
from pednstream_amd.flatten import flatten_network  # noqa: E402
from pednstream_amd.network import LINK_FIELDS  # noqa: E402

lo, hi = int(sys.argv[1]), int(sys.argv[2])
ran = skipped = 0
for seed in range(lo, hi):
    if seed > lo and (seed - lo) % 100 == 0:     # heartbeat (gpurun kills a run that is silent for 7 minutes)
        print(f"#   ... seed {seed} of {lo}..{hi}", flush=True)
    adj, params, origins, dests = random_case(seed)
    if seed % 7 == 0:
        params["assign_flows_type"] = "optimal"
    np.random.seed(seed)
    try:
        # every third network with a batch whose 128-replica segments do not halve (the chains then take 256 + 128 replicas)
        R = 384 if seed % 3 == 0 else 256
        net = Network(adj, copy.deepcopy(params), origin_nodes=origins, destination_nodes=dests, verbose=False, n_replicas=R, rng_seed=seed,
                      history="recent" if seed % 5 == 0 else "full")
    except KeyError:
        skipped += 1
        continue
    if flatten_network(net)["max_degree"] > 8:
        skipped += 1
        continue
    T = params["simulation_steps"]
    e = net.engine()
    first = 0 if seed % 5 else T - 2
    out = []
    for plan in (1, 2):
        e.set_streams(plan)
        e.reset()
        rng = np.random.default_rng(seed)
        t = 1
        while t < T:                                     # cut the run into calls of random length (short ones stay on one stream)
            n = min(int(rng.integers(3, 60)), T - t)
            net.run(t, t + n, check=False)
            t += n
        # sending / receiving flow of step t are entries t - 1: in recent-history mode their ring ends one entry earlier
        last = {f: T - 1 if first and f in ("sending_flow", "receiving_flow") else T for f in ALL_FIELDS}
        fields = {f: e.read_block(LINK_FIELDS[f][0], min(first, last[f] - 2), last[f]) for f in ALL_FIELDS}
        fields["flags"] = e.error_flags()[1]
        fields["tf"] = np.stack([np.concatenate([e.get_turning_fractions(nd.index, r) for nd in net.nodes.values()]) for r in (0, 127, 128, R - 1)])
        out.append(fields)
    for f in out[0]:
        assert np.array_equal(out[0][f], out[1][f]), (seed, f)
    ran += 1
    net.close()
print(f"two chains of launches == one chain: {ran} random networks x 256 / 384 replicas bit-exact in every field, turning fractions and flags "
      f"(every 7th with the node LP, every 5th in recent-history mode), {skipped} networks skipped (KeyError like the reference)")
