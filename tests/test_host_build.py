"""Host-side boundary: this repository's scenario loader / Network builder / route tables against the
structures captured from the reference (tests/golden/*.npz, written by oracle/gen_golden.py)."""
import json

import numpy as np
import pytest

from golden_util import Golden, build_network
from pednstream_amd.flatten import flatten_network

CASES = ["six_node_full", "nine_full", "long_corridor_full", "small_network_full", "i45_prefix", "delft_prefix",
         "melbourne_prefix", "forky", "odd_params", "odd_separators", "star8", "edge_window_gt_T", "edge_empty", "rand_nine_a", "rand_nine_b", "rand_delft_a",
         "rand_delft_b", "randnet_i45_a", "randnet_i45_b", "randnet_nine"]


@pytest.mark.parametrize("case", CASES)
def test_topology_params_and_tables_match_reference(case):
    g = Golden(case)
    net = build_network(g)
    links = list(net.links.values())
    assert [list(k) for k in net.links.keys()] == g.static("link_uv").tolist()
    for name, attr in (("length", "length"), ("width", "_width"), ("free_flow_speed", "free_flow_speed"),
                       ("k_critical", "k_critical"), ("k_jam", "k_jam"), ("gamma", "gamma"),
                       ("activity_probability", "activity_probability"), ("bi_factor", "bi_factor"),
                       ("noise_std", "speed_noise_std")):
        assert np.array_equal(np.array([float(getattr(l, attr)) for l in links]), g.static("link_" + name)), name
    assert [l.fd_type for l in links] == g.static("link_fd_type").tolist()
    assert [int(l.is_separator) for l in links] == g.static("link_is_separator").tolist()
    assert [l.free_flow_tau for l in links] == g.static("link_free_flow_tau").tolist()
    assert [l.tau_shockwave for l in links] == g.static("link_tau_sw").tolist()
    assert [l.avg_travel_time_window for l in links] == g.static("link_window").tolist()
    assert np.array_equal(np.array([l.travel_time0 for l in links], dtype=np.float32), g.static("link_tt0"))

    for ref_node, (nid, node) in zip(g.meta["nodes"], net.nodes.items()):
        assert ref_node["id"] == nid and ref_node["kind"] == node.kind
        assert ref_node["incoming"] == [l.link_id for l in node.incoming_links]
        assert ref_node["outgoing"] == [l.link_id for l in node.outgoing_links]
        assert ref_node["virtual"] == (node.virtual_incoming_link is not None)

    pf = net.path_finder
    if "od_paths" in g.meta:
        assert {f"{o}_{d}": p for (o, d), p in pf.od_paths.items()} == g.meta["od_paths"]
        assert [f"{o}_{d}" for (o, d) in pf.od_paths] == list(g.meta["od_paths"].keys())
        assert sorted(pf.nodes_in_paths) == g.meta["nodes_in_paths"]
        ref_pf = g.meta["path_finder"]
        assert (pf.temp, pf.alpha, pf.beta, pf.omega, pf.epsilon, pf.k_paths) == tuple(
            ref_pf[k] for k in ("temp", "alpha", "beta", "omega", "epsilon", "k_paths"))
        assert np.array_equal(net.od_manager.as_matrix(), g.static("od_flows"))
        assert set(int(k) for k in g.meta["turn_tables"]) == set(pf.tables.keys())
        for nid, ref_tbl in g.meta["turn_tables"].items():
            t = pf.tables[int(nid)]
            td = [[[o, d], [[up, [[dn, float(dist)] for dn, dist in downs.items()]] for up, downs in ups.items()]]
                  for (o, d), ups in t.turns_distances.items()]
            assert td == ref_tbl["turns_distances"]
            assert [[up, [[o, d] for (o, d) in ods]] for up, ods in t.up_od_probs.items()] == ref_tbl["up_od_probs"]
            assert [[[up, dn], [[o, d] for (o, d) in ods]] for (up, dn), ods in t.ods_in_turns.items()] == ref_tbl["ods_in_turns"]
    else:
        assert pf is None


def test_demand_generator_reproduces_seeded_reference_demand():
    """Config #1 KAT: yaml seed 42 -> node-1 demand, sum 6735."""
    with open("tests/golden/kat_six_node.json") as f:
        kat = json.load(f)
    from pednstream_amd import NetworkEnvGenerator
    from golden_util import DATA

    net = NetworkEnvGenerator(DATA).create_network("od_flow_example", verbose=False)
    d = np.asarray(net.nodes[1].demand, dtype=float)
    assert d[:8].tolist() == [5, 4, 4, 5, 5, 3, 5, 4]
    assert d.sum() == 6735
    assert d.tolist() == kat["plain"]["demand_node1"]


def test_flatten_shapes_and_slot_pairing():
    g = Golden("melbourne_prefix")
    net = build_network(g)
    m = flatten_network(net)
    assert (m["n_nodes"], m["n_links"], m["n_turns"], m["T"]) == (341, 938, 2106, 500)
    assert m["max_degree"] <= 8
    # slot k pairs the two directions of one corridor
    L = m["n_links"]
    for a, b in zip(m["slot_in_link"], m["slot_out_link"]):
        if a < L:
            assert m["link_rev"][a] == b
        else:
            assert b == a + 1
    # every physical link is incoming at exactly one slot and outgoing at exactly one slot
    assert sorted(x for x in m["slot_in_link"] if x < L) == list(range(L))
    assert sorted(x for x in m["slot_out_link"] if x < L) == list(range(L))


def test_gate_and_separator_setters_mirror_to_reverse_link():
    g = Golden("long_corridor_full")
    net = build_network(g)
    a, b = net.links[(2, 3)], net.links[(3, 2)]
    assert a.is_separator and b.is_separator and a.separator_width == 2.0
    a.separator_width = 1.25
    assert (a.front_gate_width, a.back_gate_width, a.separator_width) == (1.25, 1.25, 1.25)
    assert (b.front_gate_width, b.back_gate_width, b.separator_width) == (2.75, 2.75, 2.75)
    c, d = net.links[(0, 1)], net.links[(1, 0)]
    c.back_gate_width = c.back_gate_width - 0.5
    assert d.front_gate_width == c.back_gate_width == 3.5 and d.back_gate_width == 4
    with pytest.raises(AttributeError):
        c.separator_width


def test_missing_scenario_raises_file_not_found():
    from pednstream_amd import NetworkEnvGenerator
    from golden_util import DATA

    with pytest.raises(FileNotFoundError):
        NetworkEnvGenerator(DATA).create_network("no_such_scenario")


def test_unknown_node_model_is_rejected():
    g = Golden("forky")
    from pednstream_amd import Network

    p = dict(g.info["params"], assign_flows_type="best")
    with pytest.raises(ValueError):                       # node.py:302
        Network(np.array(g.info["adjacency"]), p, origin_nodes=[0, 4], verbose=False)
    Network(np.array(g.info["adjacency"]), dict(g.info["params"], assign_flows_type="optimal"), origin_nodes=[0, 4], verbose=False)


def test_array_statics_equal_the_scalar_expressions():
    """derive_statics_arrays (batched scenario draws) == derive_statics (link.py:58-63,83-86,380) element by element."""
    from pednstream_amd.scenarios import derive_statics, derive_statics_arrays

    rng = np.random.default_rng(0)
    for unit_time in (10, 5.0, 20):
        L, R = 60, 20
        length = rng.uniform(5, 400, (L, 1))
        vf = rng.uniform(0.5, 1.6, (L, R))
        kc = rng.uniform(0.5, 3, (L, R))
        kj = kc * rng.uniform(2.0, 4.0, (L, R))
        tt0, fft, tsw = derive_statics_arrays(length, vf, kc, kj, unit_time)
        for i in range(L):
            for j in range(R):
                assert derive_statics(float(length[i, 0]), vf[i, j], kc[i, j], kj[i, j], unit_time) == (tt0[i, j], fft[i, j], tsw[i, j])


def test_measured_packing_cost_travels_from_the_scenario_directory_to_the_model_description():
    """data/<scenario>/pack_cost.json (tools/pack_calibrate.py) -> Network.node_pack_cost -> flatten_network()['node_cost'] in node
    order -> pedn_model_desc.node_cost; a scenario without the file hands over NULL (the static estimate packs the bins)."""
    from golden_util import DATA
    from pednstream_amd import NetworkEnvGenerator
    from pednstream_amd.engine import build_model_desc

    np.random.seed(0)
    net = NetworkEnvGenerator(DATA).create_network("melbourne", verbose=False)
    with open(f"{DATA}/melbourne/pack_cost.json") as f:
        ref = {int(k): v for k, v in json.load(f)["node_cost"].items()}
    m = flatten_network(net)
    assert m["node_cost"].dtype == np.float32 and len(m["node_cost"]) == m["n_nodes"] == len(ref)
    assert all(np.float32(ref[int(nid)]) == c for nid, c in zip(m["node_id"], m["node_cost"])) and m["node_cost"].min() > 0
    desc, keep = build_model_desc(m)
    assert bool(desc.node_cost) and desc.node_cost[0] == m["node_cost"][0]
    np.random.seed(0)
    six = NetworkEnvGenerator(DATA).create_network("od_flow_example", verbose=False)
    m6 = flatten_network(six)
    assert m6["node_cost"] is None and not bool(build_model_desc(m6)[0].node_cost)
