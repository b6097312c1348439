"""Helpers shared by the parity tests: load a golden fixture, rebuild the same scenario with this repository's own
host code, and drive the CPU oracle / HIP engine through the same sequence of steps and mutations."""
import json
import os

import numpy as np

from pednstream_amd import Network, NetworkEnvGenerator
from pednstream_amd.flatten import flatten_network

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "data")

F64_FIELDS = ["inflow", "outflow", "cumulative_inflow", "cumulative_outflow", "sending_flow", "receiving_flow",
              "back_gate_width_data"]
F32_FIELDS = ["travel_time", "avg_travel_time", "num_pedestrians", "density", "speed", "link_flow"]
ALL_FIELDS = F64_FIELDS + F32_FIELDS


class Golden:
    def __init__(self, case):
        self.case = case
        z = np.load(os.path.join(GOLDEN, case + ".npz"), allow_pickle=False)
        self.z = z
        self.info = json.loads(str(z["info_json"]))
        self.meta = json.loads(str(z["static_meta_json"]))
        self.steps = self.info["steps_run"]          # histories are valid for t < steps
        self.seed, self.replica, self.mode = self.info["seed"], self.info["replica"], self.info["mode"]
        self.mutations = [tuple(m) for m in self.info.get("mutations", [])]

    def state(self, name):
        return self.z["state_" + name]

    def static(self, name):
        return self.z["static_" + name]

    def demand(self):
        return {int(k[len("static_demand_"):]): self.z[k] for k in self.z.files if k.startswith("static_demand_")}


def _demand_callables(g: Golden):
    """The callables a fixture names (tests/demand_callables.py), or None."""
    names = g.info.get("demand_callables")
    if not names:
        return None
    import demand_callables as dc

    made = {"plateau_pattern": dc.plateau_pattern, "node_demand_from_table": dc.make_table_demand(dc.MELBOURNE_TABLE)}
    return [made[n] for n in names]


def build_network(g: Golden, **kw):
    """This repository's Network for the golden's scenario, with the golden's demand arrays injected -- except where the
    fixture was built with custom demand callables: there the demand must come out of the callables, through the boundary."""
    np.random.seed(g.info["np_seed"])
    fns = _demand_callables(g)
    if fns is not None:
        if g.info["scenario"] is not None:
            net = NetworkEnvGenerator(DATA).create_network(g.info["scenario"], fns, demand_params_overrides=g.info.get("demand_params_overrides"),
                                                           verbose=False, **kw)
        else:
            net = Network(np.array(g.info["adjacency"]), g.info["params"], origin_nodes=g.info["origin_nodes"],
                          destination_nodes=g.info["destination_nodes"], demand_pattern=fns, verbose=False, **kw)
            if g.info["tf_nodes"]:
                net.update_turning_fractions_per_node(g.info["tf_nodes"], np.array(g.info["tf_values"]))
        for nid, arr in g.demand().items():      # what the reference's Network got out of the same callables
            mine = np.asarray(net.nodes[nid].demand, dtype=np.float64)
            assert mine.shape == arr.shape and np.array_equal(mine, arr), f"demand of node {nid} differs from the reference's"
        return net
    if g.info.get("randomize_network_seed") is not None:
        # the reference called create_network(name) and then randomize_network(name, seed) on the same generator
        gen = NetworkEnvGenerator(DATA)
        gen.create_network(g.info["scenario"], verbose=False)
        net = gen.randomize_network(g.info["scenario"], seed=g.info["randomize_network_seed"], verbose=False, **kw)
    elif g.info.get("randomized"):
        # the reference built this scenario with create_network(name, od_flows, link_params_overrides, demand_params_overrides)
        rz = g.info["randomized"]
        gen = NetworkEnvGenerator(DATA)
        T = gen.create_network(g.info["scenario"], verbose=False).simulation_steps
        od = {tuple(int(x) for x in k.split("_")): np.full(T + 1, w) for k, w in rz["od_flows"].items()}
        net = gen.create_network(g.info["scenario"], od_flows=od, link_params_overrides=rz["link_params_overrides"],
                                 demand_params_overrides=rz["demand_params_overrides"], verbose=False, **kw)
    elif g.info["scenario"] is not None:
        net = NetworkEnvGenerator(DATA).create_network(g.info["scenario"], verbose=False, **kw)
    else:
        net = Network(np.array(g.info["adjacency"]), g.info["params"], origin_nodes=g.info["origin_nodes"],
                      destination_nodes=g.info["destination_nodes"], verbose=False, **kw)
        if g.info["tf_nodes"]:
            net.update_turning_fractions_per_node(g.info["tf_nodes"], np.array(g.info["tf_values"]))
    for nid, arr in g.demand().items():
        net.nodes[nid].demand = arr
    for lid, (attr, w) in g.info.get("rl", {}).get("init_widths", {}).items():   # widths set before the first env step
        setattr(net.links[tuple(int(x) for x in lid.split("_"))], attr, w)
    return net


def apply_mutation(net, mut):
    """Apply one recorded between-step mutation through the link views (exercises the setter mirroring)."""
    _, kind, u, v, val = mut
    link = net.links[(u, v)]
    if kind == "back_gate_delta":
        link.back_gate_width = link.back_gate_width + val
    elif kind == "back_gate_set":
        link.back_gate_width = val
    elif kind == "front_gate_set":
        link.front_gate_width = val
    elif kind == "separator_set":
        link.separator_width = val
    else:
        raise ValueError(kind)


def run_oracle(g: Golden, net=None, steps=None):
    """Run the CPU oracle over the golden's steps; returns (oracle, tf history [steps-1, n_turns])."""
    import oracle_driver as od

    net = net or build_network(g)
    model = flatten_network(net)
    o = od.Oracle(model, seed=g.seed, replica=g.replica, mode=g.mode)
    for node in net.nodes.values():
        tf = net._tf_host.get(node.index)
        if tf is not None:
            o.set_tf(node.index, tf[:, 0])
    last = g.steps if steps is None else min(steps, g.steps)
    tfh = []
    for t in range(0, last):
        if t > 0:
            o.step(t)
            tfh.append(o.tf())
        for mut in g.mutations:          # t = 0: changes made before the first step
            if mut[0] == t:
                apply_mutation(net, mut)
                for which, code in (("front", 0), ("back", 1), ("sep", 2), ("sepnp", 3)):
                    for l in range(model["n_links"]):
                        o.set_width(code, l, net._widths[which][l, 0])
    return o, np.array(tfh), model, net


def compare_fields(get_field, g: Golden, n_links, last, f32_exact=True):
    """get_field(name) -> [columns, >=last].  Returns a list of mismatch descriptions (empty = parity)."""
    problems = []
    for name in ALL_FIELDS:
        mine = np.asarray(get_field(name))[:n_links, :last]
        ref = g.state(name)[:, :last]
        if mine.dtype != ref.dtype:
            problems.append(f"{name}: dtype {mine.dtype} != {ref.dtype}")
            continue
        if name in F64_FIELDS or f32_exact:
            bad = mine != ref
        else:
            bad = np.abs(mine.astype(np.float64) - ref) > 1e-6 * np.maximum(np.abs(ref), 1e-30)
        if bad.any():
            idx = np.argwhere(bad)
            t0 = idx[:, 1].min()
            l0 = idx[idx[:, 1] == t0][0][0]
            problems.append(f"{name}: {bad.sum()} mismatches, first at t={t0} link={l0}: {mine[l0, t0]!r} vs {ref[l0, t0]!r}")
    return problems


DIGEST_CASES = ["melbourne_full", "melbourne_heavy_a", "melbourne_heavy_b", "delft_full", "i45_full", "two_coordinators_full"]


def step_digests(arr):
    """uint64 [T]: blake2b-8 of the bytes of arr[:, t] (all columns) for every time index -- as oracle/gen_golden.py stores them."""
    import hashlib

    a = np.ascontiguousarray(np.asarray(arr).T)
    return np.array([int.from_bytes(hashlib.blake2b(a[t].tobytes(), digest_size=8).digest(), "little") for t in range(a.shape[0])],
                    dtype=np.uint64)


def compare_digests(get_field, g: Golden, n_links, tf_hist=None):
    """Full-horizon goldens (oracle/gen_golden.py: digest_case): every array at every time index through its digest over
    all links, the complete arrays of the sampled links, the virtual links' flows and the turning-fraction digests."""
    problems = []
    last = g.steps
    sample = g.state("sample_links")
    for name in ALL_FIELDS:
        mine = np.asarray(get_field(name))
        ref_small = g.state(name)[:, :last]
        got_small = mine[sample, :last]
        if got_small.dtype != ref_small.dtype:
            problems.append(f"{name}: dtype {got_small.dtype} != {ref_small.dtype}")
            continue
        d = step_digests(mine[:n_links, :last])
        bad = np.flatnonzero(d != g.state("digest_" + name)[:last])
        if len(bad):
            problems.append(f"{name}: digest over all links differs at {len(bad)} time indices, first t={bad[0]}")
        if not np.array_equal(got_small, ref_small):
            idx = np.argwhere(got_small != ref_small)
            t0 = idx[:, 1].min()
            l0 = idx[idx[:, 1] == t0][0][0]
            problems.append(f"{name}: sampled link {sample[l0]} differs first at t={t0}: {got_small[l0, t0]!r} vs {ref_small[l0, t0]!r}")
    for tag, off in (("vin", 0), ("vout", 1)):
        for name in ("inflow", "outflow", "cumulative_inflow", "cumulative_outflow"):
            key = f"{tag}_{name}"
            if "state_" + key in g.z.files:
                mine = np.asarray(get_field(name))[n_links + off::2, :last]
                if not np.array_equal(mine, g.state(key)[:, :last]):
                    problems.append(f"{key}: virtual links differ")
    if tf_hist is not None:
        d = step_digests(np.asarray(tf_hist).T)
        bad = np.flatnonzero(d != g.state("tf_digest"))
        if len(bad):
            problems.append(f"turning fractions: digest differs at {len(bad)} steps, first index {bad[0]}")
    return problems
