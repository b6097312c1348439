"""OutputHandler (reference on-disk format): JSON produced from this repository's run must equal what the reference's own
handlers/output_handler.py wrote for the same run (texts captured in tests/golden/output_*.npz)."""
import json
import zlib
from types import SimpleNamespace

import pytest

from golden_util import Golden, build_network, run_oracle
from pednstream_amd.output_handler import OutputHandler

CASES = ["output_six_node", "output_corridor"]


def ref_json(g, name):
    return json.loads(zlib.decompress(g.z["json_" + name].tobytes()).decode())


def check(handler_dicts, g):
    link_data, node_data, params = handler_dicts
    assert json.loads(json.dumps(link_data)) == ref_json(g, "link_data")
    assert json.loads(json.dumps(node_data)) == ref_json(g, "node_data")
    assert json.loads(json.dumps(params)) == ref_json(g, "network_params")
    assert list(link_data.keys()) == list(ref_json(g, "link_data").keys())


@pytest.mark.parametrize("case", CASES)
def test_json_from_cpu_oracle_histories_equals_reference(case, tmp_path):
    g = Golden(case)
    o, _, model, net = run_oracle(g)
    # a network-like object whose links carry plain arrays (the handler only needs the reference's attribute surface)
    links = {}
    for key, lk in net.links.items():
        ns = SimpleNamespace(link_id=lk.link_id, length=lk.length, width=lk.width, free_flow_speed=lk.free_flow_speed,
                             k_critical=lk.k_critical, k_jam=lk.k_jam, is_separator=lk.is_separator)
        for name in ("density", "link_flow", "speed", "travel_time", "inflow", "outflow", "num_pedestrians", "cumulative_inflow",
                     "cumulative_outflow", "sending_flow", "receiving_flow", "back_gate_width_data"):
            setattr(ns, name, o.field(name)[lk.index])
        if lk.is_separator:
            rec = o.field("back_gate_width_data")[lk.index].copy()
            rec[0] = lk._width / 2
            rec[g.steps:] = lk._width / 2
            ns.separator_width_data = rec
        links[key] = ns
    fake = SimpleNamespace(links=links, nodes=net.nodes, simulation_steps=net.simulation_steps, unit_time=net.unit_time,
                           destination_nodes=net.destination_nodes, origin_nodes=net.origin_nodes, path_finder=net.path_finder,
                           controller_gaters=net.controller_gaters, n_links=len(links))
    h = OutputHandler(base_dir=str(tmp_path), simulation_dir="run")
    check(h.build_network_state(fake), g)
    h.save_network_state(fake)
    loaded = OutputHandler.load_simulation(str(tmp_path / "run"))
    assert loaded["link_data"] == ref_json(g, "link_data")


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_json_from_engine_equals_reference(case, tmp_path):
    from golden_util import apply_mutation

    g = Golden(case)
    net = build_network(g, n_replicas=2, replica_offset=g.replica, rng_seed=g.seed)
    for t in range(1, g.steps):
        net.network_loading(t)
        for mut in g.mutations:
            if mut[0] == t:
                apply_mutation(net, mut)
    h = OutputHandler(base_dir=str(tmp_path), simulation_dir="run")
    check(h.build_network_state(net, replica=0), g)
    h.save_time_series(net)
    assert (tmp_path / "run" / "time_series.csv").exists()
    net.close()
