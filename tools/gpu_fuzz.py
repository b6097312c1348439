"""GPU-vs-oracle fuzz campaign beyond the 40 networks of tests/test_gpu_parity.py: random scenarios from tests/fuzz_cases.py
(the generator of the offline reference campaign oracle/fuzz_vs_reference.py), 3 replicas each, every field and the turning
fractions bit for bit, sticky error flags equal.

    python tools/gpu_fuzz.py 4000 4300            # seeds; PEDN_FUSE_TP / PEDN_LINK_OWNER / PEDN_INLINE_TF select the launch plan
    python tools/gpu_fuzz.py 4000 4300 scenarios  # additionally every replica gets its own k_critical / k_jam / free-flow speed,
                                                  # OD weights and demand (ScenarioBatch), the oracle the same per replica
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
per_replica = len(sys.argv) > 3 and sys.argv[3] == "scenarios"
from pednstream_amd.scenarios import ScenarioBatch, derive_statics_arrays
ran = skipped = flagged = 0
for seed in range(lo, hi):
    if seed > lo and (seed - lo) % 200 == 0:     # heartbeat: a long campaign keeps writing (gpurun kills a run that is silent for 7 minutes)
        print(f"#   ... seed {seed} of {lo}..{hi}: {ran} replica runs bit-exact so far", flush=True)
    adj, params, origins, dests = random_case(seed)
    if os.environ.get("PEDN_FUZZ_OPTIMAL"):          # the node LP instead of the classic rule (engine and oracle: the same simplex)
        params["assign_flows_type"] = "optimal"
    np.random.seed(seed)
    try:
        net = Network(adj, copy.deepcopy(params), origin_nodes=origins, destination_nodes=dests, verbose=False, n_replicas=3, rng_seed=seed, replica_offset=seed % 5)
    except KeyError:
        skipped += 1; continue
    model = flatten_network(net)
    if model["max_degree"] > 8:              # kernel limit (PEDN_MAX_DEGREE): pedn_create refuses such a junction
        skipped += 1; continue
    T = params["simulation_steps"]
    models = [model] * 3
    if per_replica:
        rng = np.random.default_rng(seed)
        b = ScenarioBatch(net)
        L = net.n_links
        pair = {}
        for l in net._link_list:                       # both directions of a corridor share the perturbation
            pair.setdefault(frozenset((l.start_node.node_id, l.end_node.node_id)), rng.uniform(0.6, 1.2, (2, 3)))
        f = np.stack([pair[frozenset((l.start_node.node_id, l.end_node.node_id))] for l in net._link_list])   # [L, 2, 3]
        b.kc = np.maximum(0.5, b.kc * f[:, 0, :])
        b.kj = np.maximum(b.kc * 2.0, b.kj * f[:, 0, :])
        b.vf = b.vf * np.minimum(f[:, 1, :], 1.0)
        length = np.array([l.length for l in net._link_list])[:, None]
        b.tt0, b.fft, b.tau_sw = derive_statics_arrays(length, b.vf, b.kc, b.kj, net.unit_time)
        b.link_params_dirty = True
        if b.od_w is not None:
            b.od_w = rng.uniform(1.0, 10.0, b.od_w.shape)
            b.od_dirty = True
        dem = {}
        for node in net.nodes.values():
            if node.virtual_incoming_link is not None and node.node_id in net.origin_nodes:
                for r in range(3):
                    dem[(node.node_id, r)] = rng.poisson(rng.uniform(2, 25), T).astype(np.float64)
        b.demand = dict(dem)
        b.commit()
        models = []
        for r in range(3):
            mr = dict(model)
            mr["link_kc"], mr["link_kj"], mr["link_vf"] = b.kc[:, r].copy(), b.kj[:, r].copy(), b.vf[:, r].copy()
            mr["link_fft"], mr["link_tau_sw"], mr["link_tt0"] = b.fft[:, r].copy(), b.tau_sw[:, r].copy(), b.tt0[:, r].copy()
            if b.od_w is not None:
                mr["od_w"] = np.repeat(b.od_w[:, r:r + 1], T + 1, axis=1)
            d = np.array(model["demand"], dtype=np.float64).copy()
            for (nid, rr), arr in dem.items():
                if rr == r:
                    row = model["node_demand_row"][net.nodes[nid].index]
                    d[row, :] = 0.0
                    d[row, :T] = arr
            mr["demand"] = d
            models.append(mr)
    net.run(1, T, check=False)
    e = net._engine
    _, flags = e.error_flags()
    for r in range(3):
        o = od.Oracle(models[r], seed=seed, replica=seed % 5 + r)
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
print(f"{'per-replica scenarios, ' if per_replica else ''}fuse_tp={os.environ.get('PEDN_FUSE_TP','auto')} link_owner={os.environ.get('PEDN_LINK_OWNER','auto')} inline_tf={os.environ.get('PEDN_INLINE_TF','auto')} seeds {lo}..{hi}: {ran} replica runs bit-exact, {flagged} stopped at a reference raise site (same flag on both sides), {skipped} networks skipped (KeyError like the reference)")
