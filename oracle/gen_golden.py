"""Generate the golden fixtures under tests/golden/ from the REAL reference.  TEST INFRASTRUCTURE.

Run in the build container (needs /root/reference):   python oracle/gen_golden.py [case ...]

Every fixture is data only: inputs (scenario name or adjacency+params, demand arrays, RNG key, between-step
mutations) and expected outputs (the 13 per-link history arrays, virtual-link arrays, turning-fraction history,
setup-time tables).  Cases:
  G1  kat_six_node            native numpy RNG known-answer numbers for config #1 (+ the gate-mutation variant)
  G2  *_full                  injected-RNG full histories for the small scenarios
  G3  *_prefix                injected-RNG prefixes for 45_intersections / delft / melbourne
  G4  nine_meanfield          mean-field mode
  G5  six_node_gate, forky    between-step gate mutations and externally imposed turning fractions
  G2r nine_replicas           config #2: replicas 0..3 with per-replica demand and RNG key
"""
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness as rh  # noqa: E402  (sets NPY_DISABLE_CPU_FEATURES before numpy is imported)

import numpy as np  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def tf_matrix(network, extras):
    """[steps-1, n_turns] turning-fraction history in node order."""
    cols = [extras["tf_hist"][int(nid)] for nid in network.nodes.keys()]
    return np.concatenate(cols, axis=1)


def save(case, static, state, extras, info):
    os.makedirs(OUT, exist_ok=True)
    payload = {}
    payload.update({"static_" + k: v for k, v in static.items()})
    payload.update({"state_" + k: v for k, v in state.items()})
    if "tf" in extras:
        payload["tf_hist"] = extras["tf"]
    info = dict(info)
    info["steps_run"] = extras["steps_run"]
    info["draws"] = {str(k): int(v) for k, v in extras["draws"].items()}
    info["numpy"] = np.__version__
    payload["info_json"] = np.array(json.dumps(info))
    path = os.path.join(OUT, case + ".npz")
    np.savez_compressed(path, **payload)
    print(f"{case}: {os.path.getsize(path) / 1024:.0f} KiB, draws {info['draws']}")


def scenario_case(case, name, steps=None, seed=0, replica=0, mode="philox", np_seed=20261003, mutations=None,
                  demand_override=None, create_kwargs=None, info_extra=None):
    muts = mutations or []

    def mutate(net, t):
        for (mt, kind, u, v, val) in muts:
            if mt == t:
                if kind == "back_gate_delta":
                    net.links[(u, v)].back_gate_width += val
                elif kind == "back_gate_set":
                    net.links[(u, v)].back_gate_width = val
                elif kind == "separator_set":
                    net.links[(u, v)].separator_width = val
                else:
                    raise ValueError(kind)

    net, static, state, extras = rh.run_reference(name, steps=steps, seed=seed, replica=replica, mode=mode,
                                                  mutate=mutate if muts else None, np_seed=np_seed, record_tf=True,
                                                  demand_override=demand_override, create_kwargs=create_kwargs)
    extras["tf"] = tf_matrix(net, extras)
    save(case, static, state, extras, {"scenario": name, "seed": seed, "replica": replica, "mode": mode,
                                       "np_seed": np_seed, "mutations": muts, **(info_extra or {})})
    return net


def direct_case(case, adj, params, origin_nodes, destination_nodes=(), steps=None, seed=0, replica=0, tf_nodes=None,
                tf_values=None, mutations=None, np_seed=20261003, demand_pattern=None):
    """demand_pattern: custom demand callables handed to Network(..., demand_pattern=[...]) (network.py:88-93); the fixture
    records their names, the tests pass the same callables (tests/demand_callables.py) to this repository's Network."""
    ref = rh.load_reference()
    np.random.seed(np_seed)
    net = ref["network"].Network(np.array(adj), params, origin_nodes=list(origin_nodes),
                                 destination_nodes=list(destination_nodes), demand_pattern=demand_pattern)
    if tf_nodes:
        net.update_turning_fractions_per_node(node_ids=list(tf_nodes), new_turning_fractions=np.array(tf_values))
    muts = mutations or []

    def mutate(n, t):
        for (mt, kind, u, v, val) in muts:
            if mt == t and kind == "back_gate_set":
                n.links[(u, v)].back_gate_width = val
            elif mt == t and kind == "front_gate_set":
                n.links[(u, v)].front_gate_width = val
            elif mt == t and kind == "separator_set":
                n.links[(u, v)].separator_width = val

    net, static, state, extras = rh.run_reference(None, steps=steps, seed=seed, replica=replica, mutate=mutate,
                                                  record_tf=True, network=net)
    extras["tf"] = tf_matrix(net, extras)
    save(case, static, state, extras, {"scenario": None, "adjacency": np.array(adj).tolist(), "params": params,
                                       "origin_nodes": list(origin_nodes), "destination_nodes": list(destination_nodes),
                                       "tf_nodes": list(tf_nodes or []), "tf_values": np.array(tf_values if tf_values is not None else []).tolist(),
                                       "seed": seed, "replica": replica, "mode": "philox", "np_seed": np_seed,
                                       "mutations": muts,
                                       **({"demand_callables": [f.__name__ for f in demand_pattern]} if demand_pattern else {})})


def step_digests(arr):
    """uint64 [T]: blake2b-8 of the bytes of arr[:, t] (all columns, C order) for every time index."""
    import hashlib

    a = np.ascontiguousarray(arr.T)       # [T, columns]
    return np.array([int.from_bytes(hashlib.blake2b(a[t].tobytes(), digest_size=8).digest(), "little") for t in range(a.shape[0])], dtype=np.uint64)


def digest_case(case, name, seed=0, replica=0, np_seed=20261003, demand_override=None, n_sample=16, sample_seed=0):
    """FULL-horizon pin of a big network in a small file: for every array and every time index a digest over all links, the
    complete arrays of `n_sample` links (chosen among the busiest and at random), the turning-fraction digests per step."""
    net, static, state, extras = rh.run_reference(name, seed=seed, replica=replica, np_seed=np_seed, record_tf=True,
                                                  demand_override=demand_override)
    tf = tf_matrix(net, extras)
    L = state["inflow"].shape[0]
    busy = np.argsort(-state["cumulative_inflow"][:, -2])[:n_sample // 2]
    rng = np.random.default_rng(sample_seed)
    rest = rng.choice(np.setdiff1d(np.arange(L), busy), n_sample - len(busy), replace=False)
    sample = np.sort(np.concatenate([busy, rest]))
    dig = {k: step_digests(v) for k, v in state.items()}
    small = {k: (v[sample] if v.shape[0] == L and not k.startswith("v") else v) for k, v in state.items() if not k.startswith("v")}
    small.update({k: v for k, v in state.items() if k.startswith("v")})           # virtual links are few: keep them whole
    payload_state = dict(small)
    payload_state.update({"digest_" + k: v for k, v in dig.items()})
    payload_state["sample_links"] = sample.astype(np.int32)
    payload_state["tf_digest"] = step_digests(tf.T)
    extras2 = {"draws": extras["draws"], "steps_run": extras["steps_run"]}
    save(case, static, payload_state, extras2, {"scenario": name, "seed": seed, "replica": replica, "mode": "philox", "np_seed": np_seed,
                                                "mutations": [], "digest": True,
                                                "totals": {"cumulative_inflow_last": float(state["cumulative_inflow"][:, extras["steps_run"] - 1].sum())}})


def randomized_case(case, name, rand_seed, steps=None, replica=0, np_seed=20261003):
    """one scenario drawn by the reference's own randomisers (generate_random_link_params / _od_flows /
    _demand_params, env_loader.py:183-259,363-424; generate_random_od_nodes is NOT applied -- it changes the topology) and
    built by the reference's create_network with those overrides.  The overrides are stored so that the other side can
    apply the same scenario to one replica of a batch."""
    ref = rh.load_reference()
    np.random.seed(np_seed)
    gen = ref["env"].NetworkEnvGenerator()
    gen.create_network(name)                                  # loads config / network_data
    link_ov = gen.generate_random_link_params(rand_seed)
    od_w = gen.generate_random_od_flows(rand_seed)
    dem_ov = gen.generate_random_demand_params(rand_seed)
    np.random.seed(np_seed + rand_seed)
    net = gen.create_network(name, od_flows=od_w, link_params_overrides=link_ov, demand_params_overrides=dem_ov)
    net, static, state, extras = rh.run_reference(None, steps=steps, seed=0, replica=replica, record_tf=True, network=net)
    extras["tf"] = tf_matrix(net, extras)
    def plain(d):
        return {k: {kk: (vv.item() if hasattr(vv, "item") else vv) for kk, vv in v.items()} for k, v in d.items()}
    save(case, static, state, extras, {"scenario": name, "seed": 0, "replica": replica, "mode": "philox", "np_seed": np_seed,
                                       "mutations": [], "randomized": {"rand_seed": rand_seed, "link_params_overrides": plain(link_ov),
                                                                       "od_flows": {f"{o}_{d}": float(w[0]) for (o, d), w in od_w.items()},
                                                                       "demand_params_overrides": plain(dem_ov)}})


def randnet_case(case, name, rand_seed, steps=None, replica=0, np_seed=20261003):
    """The reference's full randomize_network(name, seed) (env_loader.py:160-181): perturbed OD NODES (topology of the
    virtual links changes), link parameters, OD weights and demand parameters, then create_network."""
    ref = rh.load_reference()
    np.random.seed(np_seed)
    gen = ref["env"].NetworkEnvGenerator()
    gen.create_network(name)
    net = gen.randomize_network(name, seed=rand_seed)
    net, static, state, extras = rh.run_reference(None, steps=steps, seed=0, replica=replica, record_tf=True, network=net)
    extras["tf"] = tf_matrix(net, extras)
    save(case, static, state, extras, {"scenario": name, "seed": 0, "replica": replica, "mode": "philox", "np_seed": np_seed,
                                       "mutations": [], "randomize_network_seed": rand_seed})


def native_ensemble(case, name, n_runs=64, times=(100, 200, 300, 400, 498), demand_fn=None):
    """G6: the reference under its OWN numpy RNG, many seeds -> per-run densities / cumulative inflows at a few
    times plus each run's demand.  Used to check that the injected RNG contract (different stream, approximate binomial and
    normal transforms) reproduces the reference's ensemble statistics, not just itself."""
    ref = rh.load_reference()
    dens, cin, demands = [], [], {}
    for k in range(n_runs):
        np.random.seed(5000 + k)
        net = ref["env"].NetworkEnvGenerator().create_network(name)
        if demand_fn is not None:                       # e.g. a heavier demand than the yaml's: busy links, congested branches
            for nid, node in net.nodes.items():
                if node.demand is not None and nid in net.origin_nodes:
                    node.demand = demand_fn(net.simulation_steps, k)
        for t in range(1, net.simulation_steps):
            net.network_loading(t)
        links = list(net.links.values())
        dens.append(np.array([[l.density[t] for t in times] for l in links], dtype=np.float32))
        cin.append(np.array([[l.cumulative_inflow[t] for t in times] for l in links], dtype=np.float64))
        for nid, node in net.nodes.items():
            if node.demand is not None:
                demands.setdefault(int(nid), []).append(np.asarray(node.demand, dtype=np.float64))
    out = {"times": np.array(times), "density": np.array(dens), "cumulative_inflow": np.array(cin),
           "info_json": np.array(json.dumps({"scenario": name, "n_runs": n_runs, "numpy": np.__version__}))}
    for nid, arrs in demands.items():
        out[f"demand_{nid}"] = np.array(arrs)
    path = os.path.join(OUT, case + ".npz")
    np.savez_compressed(path, **out)
    print(f"{case}: {os.path.getsize(path) / 1024:.0f} KiB ({n_runs} native-RNG runs)")


def lp_case(case, name, steps=60, np_seed=20261003, max_solves=600):
    """assign_flows_type 'optimal': every linear programme the REAL reference hands to scipy/HiGHS during the first steps of a
    run (RegularNode.solve, node.py:249-271) -- its inputs (degree, s, r, turning fractions) and HiGHS's optimal value and
    resulting q -- so that the simplex of oracle/pedn_oracle.c can be checked against the reference's own solves."""
    ref = rh.load_reference()
    node_mod = ref["node"]
    real_linprog = node_mod.linprog
    last = {}

    def recording_linprog(*a, **k):
        res = real_linprog(*a, **k)
        last["res"] = res
        return res

    real_solve = node_mod.RegularNode.solve
    rec = []

    def recording_solve(self, s, r, type="classic"):
        tf = np.array(self.turning_fractions, dtype=np.float64)
        real_solve(self, s, r, type=type)
        if type == "optimal" and len(rec) < max_solves:
            res = last["res"]
            rec.append((self.source_num, np.array(s, dtype=np.float64), np.array(r, dtype=np.float64), tf, float(res.fun), bool(res.success),
                        np.array(self.q, dtype=np.float64)))

    node_mod.linprog = recording_linprog
    node_mod.RegularNode.solve = recording_solve
    try:
        np.random.seed(np_seed)
        gen = ref["env"].NetworkEnvGenerator()
        gen.network_data = gen.load_network_data(name)
        gen.config["params"]["assign_flows_type"] = "optimal"
        net = gen.create_network(name)
        for t in range(1, steps):
            net.network_loading(t)
    finally:
        node_mod.linprog = real_linprog
        node_mod.RegularNode.solve = real_solve
    M = max(x[0] for x in rec)
    n = len(rec)
    out = {"m": np.array([x[0] for x in rec], dtype=np.int32), "s": np.zeros((n, M)), "r": np.zeros((n, M)), "tf": np.zeros((n, M * (M - 1))),
           "fun": np.array([x[4] for x in rec]), "success": np.array([x[5] for x in rec]), "q": np.zeros((n, 2 * M)),
           "info_json": np.array(json.dumps({"scenario": name, "steps": steps, "numpy": np.__version__, "scipy": __import__("scipy").__version__}))}
    for k, (m, s_, r_, tf, fun, ok, q) in enumerate(rec):
        out["s"][k, :m], out["r"][k, :m], out["tf"][k, :m * (m - 1)], out["q"][k, :2 * m] = s_, r_, tf, q
    path = os.path.join(OUT, case + ".npz")
    np.savez_compressed(path, **out)
    print(f"{case}: {os.path.getsize(path) / 1024:.0f} KiB, {n} linear programmes of the reference, degrees {sorted(set(out['m'].tolist()))}, "
          f"{int((out['s'].sum(axis=1) > 0).sum())} with traffic")


def rl_case(case, name, obs_mode="option3", normalize=False, action_gap=1, env_steps=150, seed=0, replica=0,
            np_seed=20261003, action_seed=1, skip_prob=0.0, init_widths=None):
    """Config #5 caller: the reference's ActionApplier / ObservationBuilder / reward (rl/builders.py,
    rl/pz_pednet_env.py:548-581) around network_loading, driven by seeded uniform actions in [-0.5, width + 0.5]
    (so that both clip bounds and the per-step delta limit are exercised)."""
    ref = rh.load_reference()
    np.random.seed(np_seed)
    net = ref["env"].NetworkEnvGenerator().create_network(name)
    static = rh.dump_static(net)
    env = rh.RefEnvShim(net, obs_mode=obs_mode, normalize_obs=normalize, action_gap=action_gap)
    am = env.agent_manager
    agents = env.possible_agents
    spec = []
    for a in agents:
        if am.get_agent_type(a) == "sep":
            f, r = am.get_separator_links(a)
            spec.append({"id": a, "type": "sep", "links": [f.link_id, r.link_id]})
        else:
            spec.append({"id": a, "type": "gate", "links": [l.link_id for l in am.get_gater_outgoing_links(a)]})
    rng = np.random.default_rng(action_seed)
    acts, obs_l, rew_l, term_l = [], [], [], []
    for lid, (attr, w) in (init_widths or {}).items():    # e.g. a separator width outside ActionApplier's clip band
        setattr(net.links[tuple(int(x) for x in lid.split("_"))], attr, w)
    with rh.InjectedRNG(net, seed=seed, replica=replica) as inj:
        for _ in range(env_steps):
            actions = {}
            row = []
            for sp in spec:
                if sp["type"] == "sep":
                    w = net.links[tuple(int(x) for x in sp["links"][0].split("_"))].width
                    a = rng.uniform(-0.5, w + 0.5, size=1).astype(np.float32)
                else:
                    ws = [net.links[tuple(int(x) for x in l.split("_"))].width for l in sp["links"]]
                    a = np.array([rng.uniform(-0.5, w + 0.5) for w in ws], dtype=np.float32)
                if skip_prob and rng.uniform() < skip_prob:     # partial action dict: this agent is not given an action (NaN row)
                    row.extend([float("nan")] * len(a))
                    continue
                actions[sp["id"]] = a
                row.extend(a.tolist())
            obs, rew, term = env.step(actions)
            acts.append(row)
            obs_l.append(np.concatenate([np.asarray(obs[a], dtype=np.float32) for a in agents]))
            rew_l.append([np.float32(rew[a]) for a in agents])
            term_l.append(bool(list(term.values())[0]))
    state = rh.dump_state(net, steps=env.sim_step)
    extras = {"draws": dict(inj.draws), "steps_run": env.sim_step}
    info = {"scenario": name, "seed": seed, "replica": replica, "mode": "philox", "np_seed": np_seed, "mutations": [],
            "rl": {"obs_mode": obs_mode, "normalize": normalize, "action_gap": action_gap, "env_steps": env_steps, "agents": spec,
                   "init_widths": init_widths or {}}}
    state["rl_actions"] = np.array(acts, dtype=np.float32)
    state["rl_obs"] = np.array(obs_l, dtype=np.float32)
    state["rl_rewards"] = np.array(rew_l, dtype=np.float32)
    state["rl_terminated"] = np.array(term_l)
    save(case, static, state, extras, info)


def output_case(case, name, mutations=None, np_seed=20261003):
    """what the reference's own OutputHandler.save_network_state writes for a run (the JSON texts are
    stored zlib-compressed; inputs as in scenario_case)."""
    import tempfile
    import zlib

    net = scenario_case(case, name, mutations=mutations, np_seed=np_seed)
    sys.path.insert(0, rh.REF_ROOT)
    from handlers.output_handler import OutputHandler

    with tempfile.TemporaryDirectory() as tmp:
        OutputHandler(base_dir=tmp, simulation_dir="run").save_network_state(net)
        texts = {f: open(os.path.join(tmp, "run", f + ".json"), "rb").read() for f in ("link_data", "node_data", "network_params")}
    path = os.path.join(OUT, case + ".npz")
    z = dict(np.load(path))
    for k, v in texts.items():
        z["json_" + k] = np.frombuffer(zlib.compress(v, 9), dtype=np.uint8)
    np.savez_compressed(path, **z)
    print(f"{case}: + reference OutputHandler JSON ({sum(len(v) for v in texts.values()) / 1024:.0f} KiB raw), {os.path.getsize(path) / 1024:.0f} KiB")


def kat_native():
    """G1 / G1b: the reference under its own numpy RNG (yaml seed 42 reseeds the global stream)."""
    ref = rh.load_reference()
    out = {}
    for tag, mutate in (("plain", False), ("gate_mutation", True)):
        gen = ref["env"].NetworkEnvGenerator()
        net = gen.create_network("od_flow_example")
        for t in range(1, gen.config["params"]["simulation_steps"]):
            net.network_loading(t)
            if mutate and t in (100, 101, 102, 103, 104, 105, 106, 107, 108):   # examples/six_node.py:29-30
                net.links[(3, 5)].back_gate_width -= 0.1
        h = hashlib.sha256()
        for key in sorted(net.links.keys()):
            lk = net.links[key]
            for arr in (lk.cumulative_inflow, lk.cumulative_outflow, lk.density, lk.speed, lk.sending_flow, lk.receiving_flow):
                h.update(np.ascontiguousarray(arr).tobytes())
        out[tag] = {
            "sum_cum_in_499": float(sum(l.cumulative_inflow[499] for l in net.links.values())),
            "link_1_3_cum_in_499": float(net.links[(1, 3)].cumulative_inflow[499]),
            "link_1_3_density_499": float(net.links[(1, 3)].density[499]),
            "link_3_5_cum_in_499": float(net.links[(3, 5)].cumulative_inflow[499]),
            "link_3_5_density_499": float(net.links[(3, 5)].density[499]),
            "link_3_5_back_gate_width": float(net.links[(3, 5)].back_gate_width),
            "link_5_3_front_gate_width": float(net.links[(5, 3)].front_gate_width),
            "demand_node1": [float(x) for x in net.nodes[1].demand],
            "sha256": h.hexdigest(),
        }
    out["numpy"] = np.__version__
    with open(os.path.join(OUT, "kat_six_node.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("kat_six_node:", {k: (v["sum_cum_in_499"], v["sha256"][:16]) for k, v in out.items() if isinstance(v, dict)})


FORKY_ADJ = [[0, 1, 0, 0, 0], [1, 0, 1, 0, 1], [0, 1, 0, 1, 0], [0, 0, 1, 0, 0], [0, 1, 0, 0, 0]]
FORKY_PARAMS = {
    "unit_time": 10, "simulation_steps": 300, "assign_flows_type": "classic",
    "default_link": {"length": 100, "width": 3, "free_flow_speed": 1.5, "k_critical": 2, "k_jam": 6, "gamma": 0,
                     "speed_noise_std": 0.05, "fd_type": "yperman", "bi_factor": 1.2},
    "links": {"1_2": {"length": 100, "width": 1, "free_flow_speed": 1.5, "k_critical": 2, "k_jam": 6,
                      "speed_noise_std": 0.05, "fd_type": "yperman", "controller_type": "gate"},
              "2_3": {"length": 50, "width": 1, "free_flow_speed": 1.5, "k_critical": 2, "k_jam": 6,
                      "speed_noise_std": 0.05, "fd_type": "yperman"}},
    "demand": {"origin_0": {"peak_lambda": 15, "base_lambda": 5}, "origin_4": {"peak_lambda": 15, "base_lambda": 5}},
}


# Parameter sets no reference scenario uses (smulders FD, non-dyadic k_c / k_j / dt, bi_factor != 1, activity + noise
# together, per-link overrides): they pin the precision ledger of the oracle away from the "nice" yaml values.
ODD_ADJ = [[0, 1, 1, 0, 0, 0, 0], [1, 0, 1, 1, 0, 0, 0], [1, 1, 0, 1, 1, 0, 0], [0, 1, 1, 0, 1, 1, 0],
           [0, 0, 1, 1, 0, 1, 1], [0, 0, 0, 1, 1, 0, 1], [0, 0, 0, 0, 1, 1, 0]]
ODD_PARAMS = {
    "unit_time": 5, "simulation_steps": 400, "assign_flows_type": "classic", "seed": 7,
    "path_finder": {"k_paths": 3, "temp": 3.7, "alpha": 1.3, "beta": 0.7, "omega": 0.45, "std_dev": 0.05},
    "default_link": {"length": 37.3, "width": 2.5, "free_flow_speed": 1.27, "k_critical": 1.73, "k_jam": 5.4, "gamma": 0.013,
                     "speed_noise_std": 0.03, "fd_type": "smulders", "bi_factor": 1.5, "activity_probability": 0.15},
    "links": {"1_3": {"length": 61.9, "fd_type": "greenshields", "k_critical": 2.1},
              "2_4": {"length": 45.5, "width": 1.7, "fd_type": "yperman", "activity_probability": 0},
              "3_5": {"length": 29.1, "speed_noise_std": 0, "gamma": 0},
              "4_6": {"length": 88.8, "free_flow_speed": 0.93, "k_jam": 6.6}},
    "controllers": {"enabled": True, "nodes": [3]},
    "demand": {"origin_0": {"peak_lambda": 22, "base_lambda": 9}, "origin_6": {"pattern": "constant", "base_lambda": 7.5}},
}
SEP_ADJ = [[0, 1, 0, 0, 0], [1, 0, 1, 0, 0], [0, 1, 0, 1, 0], [0, 0, 1, 0, 1], [0, 0, 0, 1, 0]]
SEP_PARAMS = {
    "unit_time": 10, "simulation_steps": 300, "assign_flows_type": "classic", "seed": 3,
    "default_link": {"length": 80, "width": 3.2, "free_flow_speed": 1.2, "k_critical": 1.9, "k_jam": 5.8, "gamma": 0.02,
                     "speed_noise_std": 0.04, "fd_type": "yperman", "bi_factor": 1, "activity_probability": 0.05},
    "links": {"1_2": {"controller_type": "separator"}, "2_3": {"controller_type": "separator", "length": 55.5, "fd_type": "greenshields"}},
    "demand": {"origin_0": {"peak_lambda": 30, "base_lambda": 18}, "origin_4": {"peak_lambda": 28, "base_lambda": 16}},
}


# Edge cases: moving-average window longer than the horizon (W = 100/dt = 50 > T = 40), and a network nobody enters.
EDGE_ADJ = [[0, 1, 0, 0], [1, 0, 1, 1], [0, 1, 0, 1], [0, 1, 1, 0]]
EDGE_PARAMS_SHORT = {
    "unit_time": 2, "simulation_steps": 40, "assign_flows_type": "classic", "seed": 9,
    "path_finder": {"k_paths": 2, "temp": 5, "alpha": 1, "beta": 0.5, "omega": 0.8},
    "default_link": {"length": 12, "width": 2.5, "free_flow_speed": 1.4, "k_critical": 2, "k_jam": 6, "gamma": 0.01,
                     "speed_noise_std": 0.05, "fd_type": "yperman", "bi_factor": 1, "activity_probability": 0.2},
    "demand": {"origin_0": {"peak_lambda": 6, "base_lambda": 4}},
}
EDGE_PARAMS_EMPTY = {
    "unit_time": 10, "simulation_steps": 120, "assign_flows_type": "classic", "seed": 9,
    "default_link": {"length": 50, "width": 2, "free_flow_speed": 1.1, "k_critical": 2, "k_jam": 6, "gamma": 0.01,
                     "speed_noise_std": 0.05, "fd_type": "greenshields"},
    "demand": {"origin_0": {"pattern": "constant", "base_lambda": 0}},
}

# A hub with 7 neighbours that is also an origin: 8 slots, the largest node the kernels support (PEDN_MAX_DEGREE).
STAR_N = 15
STAR_ADJ = [[0] * STAR_N for _ in range(STAR_N)]
for _k in range(1, 8):
    STAR_ADJ[0][_k] = STAR_ADJ[_k][0] = 1                  # hub 0 -- spokes 1..7
    STAR_ADJ[_k][_k + 7] = STAR_ADJ[_k + 7][_k] = 1          # spoke k -- leaf k+7
STAR_ADJ[1][2] = STAR_ADJ[2][1] = 1                        # one chord so that several routes exist
STAR_PARAMS = {
    "unit_time": 10, "simulation_steps": 260, "assign_flows_type": "classic", "seed": 5,
    "path_finder": {"k_paths": 3, "temp": 4.0, "alpha": 1.0, "beta": 0.6, "omega": 0.7},
    "default_link": {"length": 60, "width": 2.0, "free_flow_speed": 1.3, "k_critical": 1.8, "k_jam": 5.5, "gamma": 0.01,
                     "speed_noise_std": 0.04, "fd_type": "yperman", "bi_factor": 1, "activity_probability": 0.1},
    "links": {"0_3": {"length": 45, "width": 1.2}, "0_5": {"length": 75}},
    "demand": {"origin_0": {"peak_lambda": 30, "base_lambda": 22}, "origin_8": {"peak_lambda": 25, "base_lambda": 12},
               "origin_12": {"peak_lambda": 20, "base_lambda": 15}},
}


def replica_demand(T, r, base=20.0, peak=25.0):
    """Config #2 per-replica origin demand: Poisson around the gaussian-peaks profile, numpy Generator(1000+r)."""
    t = np.arange(T)
    lam = base + peak * np.exp(-(t - T / 4) ** 2 / (2 * (T / 20) ** 2)) + peak * np.exp(-(t - 3 * T / 4) ** 2 / (2 * (T / 20) ** 2))
    return np.random.default_rng(1000 + r).poisson(lam).astype(np.float64)


CASES = {
    "kat": kat_native,
    "six_node_full": lambda: scenario_case("six_node_full", "od_flow_example"),
    "nine_full": lambda: scenario_case("nine_full", "nine_intersections"),
    "long_corridor_full": lambda: scenario_case("long_corridor_full", "long_corridor",
                                                mutations=[(150, "separator_set", 2, 3, 1.25), (300, "separator_set", 2, 3, 2.5)]),
    "small_network_full": lambda: scenario_case("small_network_full", "small_network"),
    "i45_prefix": lambda: scenario_case("i45_prefix", "45_intersections", steps=250),
    "delft_prefix": lambda: scenario_case("delft_prefix", "delft", steps=70),
    "melbourne_prefix": lambda: scenario_case("melbourne_prefix", "melbourne", steps=130),
    "nine_meanfield": lambda: scenario_case("nine_meanfield", "nine_intersections", steps=200, mode="meanfield"),
    "six_node_gate": lambda: scenario_case("six_node_gate", "od_flow_example",
                                           mutations=[(t, "back_gate_delta", 3, 5, -0.1) for t in range(100, 109)]),
    "forky": lambda: direct_case("forky", FORKY_ADJ, FORKY_PARAMS, [0, 4], tf_nodes=[1],
                                 tf_values=[[1, 0, 0.5, 0.5, 0, 1]],
                                 mutations=[(40, "back_gate_set", 1, 2, 0.0), (120, "back_gate_set", 1, 2, 1.0)]),
    "edge_window_gt_T": lambda: direct_case("edge_window_gt_T", EDGE_ADJ, EDGE_PARAMS_SHORT, [0], destination_nodes=[2, 3], seed=4, replica=1),
    "edge_empty": lambda: direct_case("edge_empty", EDGE_ADJ, EDGE_PARAMS_EMPTY, [0], seed=4, replica=0),
    "star8": lambda: direct_case("star8", STAR_ADJ, STAR_PARAMS, [0, 8, 12], destination_nodes=[9, 10, 11, 13, 14, 0], seed=21, replica=2),
    "odd_params": lambda: direct_case("odd_params", ODD_ADJ, ODD_PARAMS, [0, 6], destination_nodes=[6, 0], seed=11, replica=5),
    "odd_separators": lambda: direct_case("odd_separators", SEP_ADJ, SEP_PARAMS, [0, 4], seed=2, replica=9,
                                          mutations=[(60, "separator_set", 1, 2, 0.9), (140, "separator_set", 2, 3, 2.4),
                                                     (200, "back_gate_set", 3, 4, 1.1)]),
}
CASES.update({
    "output_six_node": lambda: output_case("output_six_node", "od_flow_example",
                                           mutations=[(t, "back_gate_delta", 3, 5, -0.1) for t in range(100, 109)]),
    "output_corridor": lambda: output_case("output_corridor", "long_corridor", mutations=[(150, "separator_set", 2, 3, 1.25)]),
    "lp_nine": lambda: lp_case("lp_nine", "nine_intersections", steps=70),
    "lp_i45": lambda: lp_case("lp_i45", "45_intersections", steps=40, max_solves=900),
    "g6_nine_native": lambda: native_ensemble("g6_nine_native", "nine_intersections"),
    "g6_melbourne_heavy_native": lambda: native_ensemble("g6_melbourne_heavy_native", "melbourne", n_runs=24,
                                                         demand_fn=lambda T, k: replica_demand(T, 300 + k, base=60.0, peak=120.0)),
    "randnet_i45_a": lambda: randnet_case("randnet_i45_a", "45_intersections", 3, steps=150),
    "randnet_i45_b": lambda: randnet_case("randnet_i45_b", "45_intersections", 8, steps=150, replica=2),
    "randnet_nine": lambda: randnet_case("randnet_nine", "nine_intersections", 5, steps=200, replica=1),
    "rand_nine_a": lambda: randomized_case("rand_nine_a", "nine_intersections", 11, steps=220, replica=0),
    "rand_nine_b": lambda: randomized_case("rand_nine_b", "nine_intersections", 12, steps=220, replica=1),
    "rand_delft_a": lambda: randomized_case("rand_delft_a", "delft", 21, steps=60, replica=0),
    "rand_delft_b": lambda: randomized_case("rand_delft_b", "delft", 22, steps=60, replica=1),
    "rl_nine_opt3": lambda: rl_case("rl_nine_opt3", "nine_intersections", obs_mode="option3", env_steps=200),
    "rl_nine_opt2n": lambda: rl_case("rl_nine_opt2n", "nine_intersections", obs_mode="option2", normalize=True, env_steps=120, action_seed=2),
    "rl_nine_opt5g2": lambda: rl_case("rl_nine_opt5g2", "nine_intersections", obs_mode="option5", action_gap=2, env_steps=90, action_seed=3),
    "rl_nine_opt4": lambda: rl_case("rl_nine_opt4", "nine_intersections", obs_mode="option4", env_steps=60, action_seed=4),
    "rl_i45_opt3": lambda: rl_case("rl_i45_opt3", "45_intersections", obs_mode="option3", env_steps=200, action_seed=5),
    "rl_i45_episode": lambda: rl_case("rl_i45_episode", "45_intersections", obs_mode="option3", env_steps=698, action_seed=15),
    "rl_corridor_opt1": lambda: rl_case("rl_corridor_opt1", "long_corridor", obs_mode="option1", env_steps=200, action_seed=6),
})
CASES.update({   # the remaining scenario directories of the reference's data/ (RL datasets with controller nodes)
    "butterfly_scA_full": lambda: scenario_case("butterfly_scA_full", "butterfly_scA", seed=3, replica=1),
    "butterfly_scB_full": lambda: scenario_case("butterfly_scB_full", "butterfly_scB", seed=3, replica=2),
    "butterfly_scC_full": lambda: scenario_case("butterfly_scC_full", "butterfly_scC", seed=3, replica=3),
    "one_intersection_full": lambda: scenario_case("one_intersection_full", "one_intersection_v0", seed=4, replica=0),
    "two_coordinators_prefix": lambda: scenario_case("two_coordinators_prefix", "two_coordinators", steps=260, seed=5, replica=4),
    "rl_nine_partial": lambda: rl_case("rl_nine_partial", "nine_intersections", obs_mode="option3", env_steps=150, action_seed=11, skip_prob=0.4),
    "rl_corridor_partial": lambda: rl_case("rl_corridor_partial", "long_corridor", obs_mode="option1", env_steps=150, action_seed=12, skip_prob=0.5,
                                           init_widths={"2_3": ("separator_width", 0.75)}),
    "rl_butterfly_opt3": lambda: rl_case("rl_butterfly_opt3", "butterfly_scC", obs_mode="option3", env_steps=200, action_seed=7),
    "rl_one_intersection_opt5": lambda: rl_case("rl_one_intersection_opt5", "one_intersection_v0", obs_mode="option5", env_steps=200, action_seed=8),
})
CASES.update({   # full-horizon pins of the two headline networks (every step, every link, through per-step digests)
    "melbourne_full": lambda: digest_case("melbourne_full", "melbourne", seed=2, replica=3),
    "i45_full": lambda: digest_case("i45_full", "45_intersections", seed=4, replica=2),
    "two_coordinators_full": lambda: digest_case("two_coordinators_full", "two_coordinators", seed=5, replica=4),
    "delft_full": lambda: digest_case("delft_full", "delft", seed=1, replica=5),
    # melbourne under heavy demand: the release binomials, the diffusion look-backs and the congested branch fire
    "melbourne_heavy_a": lambda: digest_case("melbourne_heavy_a", "melbourne", seed=2, replica=7,
                                             demand_override={289: replica_demand(500, 7, base=60.0, peak=120.0)}),
    "melbourne_heavy_b": lambda: digest_case("melbourne_heavy_b", "melbourne", seed=2, replica=8,
                                             demand_override={289: replica_demand(500, 8, base=150.0, peak=300.0)}),
})
for _r in range(4):
    CASES[f"nine_replica{_r}"] = (lambda r=_r: scenario_case(
        f"nine_replica{r}", "nine_intersections", steps=160, seed=0, replica=r,
        demand_override={0: replica_demand(500, 3 * r + 0), 8: replica_demand(500, 3 * r + 1),
                         2: replica_demand(500, 3 * r + 2, peak=50.0)}))


sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import demand_callables as dc  # noqa: E402

CASES.update({   # the shortest horizons the reference's loop `for t in range(1, simulation_steps)` can run: one step, two steps
    "edge_T2": lambda: direct_case("edge_T2", EDGE_ADJ, dict(EDGE_PARAMS_SHORT, simulation_steps=2), [0], destination_nodes=[2, 3], seed=4, replica=0),
    "edge_T3": lambda: direct_case("edge_T3", EDGE_ADJ, dict(EDGE_PARAMS_SHORT, simulation_steps=3), [0], destination_nodes=[2, 3], seed=4, replica=2),
})
CASES.update({   # the calling sequence of examples/forky_queues.py:71-118: a front gate narrowed BEFORE the first step, opened at a later one
    "forky_front": lambda: direct_case("forky_front", FORKY_ADJ, dict(FORKY_PARAMS, simulation_steps=420), [0, 4], tf_nodes=[1],
                                       tf_values=[[1, 0, 0.5, 0.5, 0, 1]], seed=8, replica=3,
                                       mutations=[(0, "front_gate_set", 1, 2, 0.5), (260, "front_gate_set", 1, 2, 3)]),
})
CASES.update({   # custom demand callables through the boundary (examples/spike.py:85-121, examples/Melbourne.py:36)
    "spike_callable": lambda: direct_case("spike_callable", dc.SPIKE_ADJ, dc.spike_params(), [4], tf_nodes=[4], tf_values=[[1, 0, 0, 1, 0, 1]],
                                          seed=6, replica=2, demand_pattern=[dc.plateau_pattern]),
    "melbourne_callable": lambda: scenario_case(
        "melbourne_callable", "melbourne", steps=170, seed=3, replica=1,
        create_kwargs={"custom_demand_functions": [dc.make_table_demand(dc.MELBOURNE_TABLE)],
                       "demand_params_overrides": {"origin_289": {"pattern": "node_demand_from_table"}}},
        info_extra={"demand_callables": ["node_demand_from_table"],
                    "demand_params_overrides": {"origin_289": {"pattern": "node_demand_from_table"}}}),
})


if __name__ == "__main__":
    todo = sys.argv[1:] or list(CASES)
    for c in todo:
        CASES[c]()
