"""GPU-vs-oracle fuzz campaign beyond the 40 networks of tests/test_gpu_parity.py: random scenarios from tests/fuzz_cases.py
(the generator of the offline reference campaign oracle/fuzz_vs_reference.py), 3 replicas each, every field and the turning
fractions bit for bit, sticky error flags equal.

    python tools/gpu_fuzz.py 4000 4300            # seeds; PEDN_FUSE_TP / PEDN_NODE_WAVES select the launch plan
"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_driver as od
from golden_util import ALL_FIELDS
from fuzz_cases import random_case
from pednstream_amd import Network
from pednstream_amd.flatten import flatten_network
from pednstream_amd.network import LINK_FIELDS
lo, hi = int(sys.argv[1]), int(sys.argv[2])
ran = skipped = flagged = 0
for seed in range(lo, hi):
    adj, params, origins, dests = random_case(seed)
    np.random.seed(seed)
    try:
        net = Network(adj, copy.deepcopy(params), origin_nodes=origins, destination_nodes=dests, verbose=False, n_replicas=3, rng_seed=seed, replica_offset=seed % 5)
    except KeyError:
        skipped += 1; continue
    model = flatten_network(net)
    T = params["simulation_steps"]
    net.run(1, T, check=False)
    e = net._engine
    _, flags = e.error_flags()
    for r in range(3):
        o = od.Oracle(model, seed=seed, replica=seed % 5 + r)
        o.run(1, T)
        assert int(flags[r]) == o.flags(), (seed, r, int(flags[r]), o.flags())
        if o.flags():
            flagged += 1; continue
        for fname in ALL_FIELDS:
            mine = e.read_block(LINK_FIELDS[fname][0], 0, T, rep0=r, rep1=r + 1)[:, :, 0].T
            assert np.array_equal(mine[:e.n_links], o.field(fname)[:e.n_links, :T]), (seed, r, fname)
        tf = np.concatenate([e.get_turning_fractions(nd.index, r) for nd in net.nodes.values()])
        assert np.array_equal(tf, o.tf()), (seed, r)
        ran += 1
    net.close()
print(f"fuse_tp={os.environ.get('PEDN_FUSE_TP','auto')} seeds {lo}..{hi}: {ran} replica runs bit-exact, {flagged} stopped at a reference raise site (same flag on both sides), {skipped} networks skipped (KeyError like the reference)")
