"""Oracle harness -- TEST INFRASTRUCTURE.  Runs the REAL reference (imported in place from
/root/reference, never copied) with the RNG contract of oracle/rng_contract.py injected at
the four hot-path draw sites, and dumps everything a parity test needs.

Only usable in the build container (the reference never travels to the GPU box); its
outputs are committed as data fixtures under tests/golden/ by oracle/gen_golden.py.

Injection points (nothing in the reference tree is modified):
  * Network.setup_logger            -> null logger (network.py:24-30,47 mkdir/open under the read-only tree)
  * Link.cal_sending_flow           -> publishes (link, t') for sites release/activity (link.py:337,343,356)
  * Link/Separator.cal_receiving_flow -> publishes (link, t') for site reverse (link.py:382)
  * Link/Separator.update_speeds    -> publishes (link, t) for site noise (functions.py:133)
  * np.random.binomial / np.random.normal -> contract transforms keyed (seed, replica, link, t, site)
"""
import json
import logging
import os
import sys

REF_ROOT = os.environ.get("PEDN_REFERENCE_ROOT", "/root/reference")

# numpy's AVX512F exp loop differs from libm's exp in the last bit for ~5% of arguments on this
# CPU; pin np.exp (path_finder.py:585) to the libm path so the goldens do not depend on the
# SIMD dispatch of the machine that generated them.  Must be set before numpy is imported.
os.environ.setdefault("NPY_DISABLE_CPU_FEATURES",
                      "AVX512F AVX512CD AVX512_SKX AVX512_CLX AVX512_CNL AVX512_ICL")
os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
os.environ.setdefault("MPLBACKEND", "Agg")
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rng_contract as rc  # noqa: E402

_loaded = {}


def load_reference():
    """Import the reference in place and neutralise its file logger."""
    if _loaded:
        return _loaded
    if not os.path.isdir(REF_ROOT):
        raise RuntimeError(f"reference tree not present at {REF_ROOT}")
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    from src.LTM import link as ref_link
    from src.LTM import network as ref_network
    from src.LTM import node as ref_node
    from src.utils import env_loader as ref_env
    from src.utils import functions as ref_functions

    def _null_logger(log_level=logging.INFO, log_dir=None):
        lg = logging.getLogger("pedn.oracle.null")
        if not lg.handlers:
            lg.addHandler(logging.NullHandler())
        lg.propagate = False
        return lg

    ref_network.Network.setup_logger = staticmethod(_null_logger)
    _loaded.update(link=ref_link, network=ref_network, node=ref_node, env=ref_env, functions=ref_functions)
    return _loaded


class InjectedRNG:
    """Context manager installing the contract RNG into the reference."""

    # caller line in link.py -> site (link.py:337,343 release; :356 activity; :382 reverse)
    _LINE_SITE = {337: rc.SITE_RELEASE, 343: rc.SITE_RELEASE, 356: rc.SITE_ACTIVITY, 382: rc.SITE_REVERSE}

    def __init__(self, network, seed=0, replica=0, mode="philox"):
        self.network = network
        self.seed, self.replica, self.mode = seed, replica, mode
        self.link_index = {lk.link_id: i for i, lk in enumerate(network.links.values())}
        self.ctx = None
        self.draws = {0: 0, 1: 0, 2: 0, 3: 0}

    def __enter__(self):
        ref = load_reference()
        Link, Separator = ref["link"].Link, ref["link"].Separator
        self._saved = []
        harness = self

        def wrap(cls, name):
            orig = cls.__dict__[name]

            def wrapped(self_link, time_step, *a, **kw):
                prev = harness.ctx
                harness.ctx = (harness.link_index[self_link.link_id], int(time_step))
                try:
                    return orig(self_link, time_step, *a, **kw)
                finally:
                    harness.ctx = prev

            self._saved.append((cls, name, orig))
            setattr(cls, name, wrapped)

        wrap(Link, "cal_sending_flow")
        wrap(Link, "cal_receiving_flow")
        wrap(Separator, "cal_receiving_flow")
        wrap(Link, "update_speeds")
        wrap(Separator, "update_speeds")

        self._binomial, self._normal = np.random.binomial, np.random.normal

        def binomial(n, p, size=None):
            assert size is None
            line = sys._getframe(1).f_lineno
            site = self._LINE_SITE[line]
            self.draws[site] += 1
            if self.mode == "meanfield":
                return rc.binomial_meanfield(n, p)
            link, t = self.ctx
            return rc.binomial(n, p, rc.Stream(self.seed, self.replica, link, t, site))

        def normal(loc=0.0, scale=1.0, size=None):
            assert size is None and loc == 0
            self.draws[rc.SITE_NOISE] += 1
            if self.mode == "meanfield":
                return 0.0
            link, t = self.ctx
            return rc.normal(scale, rc.Stream(self.seed, self.replica, link, t, rc.SITE_NOISE))

        np.random.binomial, np.random.normal = binomial, normal
        return self

    def __exit__(self, *exc):
        np.random.binomial, np.random.normal = self._binomial, self._normal
        for cls, name, orig in self._saved:
            setattr(cls, name, orig)
        return False


LINK_F64 = ("inflow", "outflow", "cumulative_inflow", "cumulative_outflow", "sending_flow", "receiving_flow",
            "back_gate_width_data")
LINK_F32 = ("travel_time", "avg_travel_time", "num_pedestrians", "density", "speed", "link_flow")
VLINK_F64 = ("inflow", "outflow", "cumulative_inflow", "cumulative_outflow")


def dump_static(network):
    """Topology, parameters and setup-time tables of a reference Network, as plain arrays/JSON."""
    ref = load_reference()
    Separator = ref["link"].Separator
    OneToOneNode = ref["node"].OneToOneNode
    out = {}
    links = list(network.links.values())
    out["link_uv"] = np.array([list(k) for k in network.links.keys()], dtype=np.int64)
    for name, attr in (("length", "length"), ("width", "_width"), ("free_flow_speed", "free_flow_speed"),
                       ("k_critical", "k_critical"), ("k_jam", "k_jam"), ("gamma", "gamma"),
                       ("activity_probability", "activity_probability")):
        out["link_" + name] = np.array([float(getattr(l, attr)) for l in links], dtype=np.float64)
    out["link_bi_factor"] = np.array([float(l.speed_density_fd.bi_factor) for l in links])
    out["link_noise_std"] = np.array([float(l.speed_density_fd.noise_std) for l in links])
    out["link_fd_type"] = np.array([l.speed_density_fd.model_type for l in links])
    out["link_is_separator"] = np.array([isinstance(l, Separator) for l in links], dtype=np.int8)
    out["link_free_flow_tau"] = np.array([int(l.free_flow_tau) for l in links], dtype=np.int64)
    out["link_window"] = np.array([int(l.avg_travel_time_window) for l in links], dtype=np.int64)
    out["link_tt0"] = np.array([l.travel_time[0] for l in links], dtype=np.float32)
    out["link_tau_sw"] = np.array([round(l.length / (l.shockwave_speed * l.unit_time)) for l in links], dtype=np.int64)

    nodes = []
    demand = {}
    for nid, node in network.nodes.items():
        def lid(l):
            return l.link_id
        nodes.append({
            "id": int(nid),
            "kind": "one_to_one" if isinstance(node, OneToOneNode) else "regular",
            "incoming": [lid(l) for l in node.incoming_links],
            "outgoing": [lid(l) for l in node.outgoing_links],
            "virtual": node.virtual_incoming_link is not None,
        })
        if node.demand is not None:
            demand[str(int(nid))] = np.asarray(node.demand, dtype=np.float64)
    tables = {}
    pf = network.path_finder
    meta = {
        "simulation_steps": int(network.simulation_steps), "unit_time": float(network.unit_time),
        "origin_nodes": [int(x) for x in network.origin_nodes],
        "destination_nodes": [int(x) for x in network.destination_nodes],
        "nodes": nodes,
    }
    if pf is not None:
        meta["path_finder"] = {"temp": float(pf.temp), "alpha": float(pf.alpha), "beta": float(pf.beta),
                               "omega": float(pf.omega), "epsilon": float(pf.epsilon), "k_paths": int(pf.k_paths)}
        meta["od_pairs"] = [[int(o), int(d)] for (o, d) in network.od_manager.od_flows.keys()]
        out["od_flows"] = np.array([np.asarray(v, dtype=np.float64) for v in network.od_manager.od_flows.values()])
        meta["od_paths"] = {f"{o}_{d}": [[int(x) for x in p] for p in paths] for (o, d), paths in pf.od_paths.items()}
        meta["nodes_in_paths"] = sorted(int(x) for x in pf.nodes_in_paths)
        for nid, node in network.nodes.items():
            if not hasattr(node, "turns_distances"):
                continue
            tables[str(int(nid))] = {
                # iteration orders are part of the contract: they fix the floating-point summation order
                "turns_distances": [[[int(o), int(d)], [[int(up), [[int(dn), float(dist)] for dn, dist in downs.items()]]
                                                        for up, downs in ups.items()]]
                                    for (o, d), ups in node.turns_distances.items()],
                "up_od_probs": [[int(up), [[int(o), int(d)] for (o, d) in ods.keys()]] for up, ods in node.up_od_probs.items()],
                "ods_in_turns": [[[int(up), int(dn)], [[int(o), int(d)] for (o, d) in ods]]
                                 for (up, dn), ods in node.ods_in_turns.items()],
            }
    meta["turn_tables"] = tables
    out["meta_json"] = np.array(json.dumps(meta))
    for k, v in demand.items():
        out["demand_" + k] = v
    return out


def dump_state(network, steps=None):
    """The 13 per-link history arrays (+4 per virtual link) as [L, T+1] stacks."""
    links = list(network.links.values())
    sl = slice(None) if steps is None else slice(0, steps)
    out = {}
    for name in LINK_F64:
        out[name] = np.stack([np.asarray(getattr(l, name), dtype=np.float64)[sl] for l in links])
    for name in LINK_F32:
        arr = [getattr(l, name) for l in links]
        assert all(a.dtype == np.float32 for a in arr), name
        out[name] = np.stack([a[sl] for a in arr])
    sep = [getattr(l, "separator_width_data", None) for l in links]
    if any(s is not None for s in sep):
        out["separator_width_data"] = np.stack([np.asarray(s if s is not None else np.zeros(len(links[0].inflow)))[sl]
                                                for s in sep])
    vin, vout = [], []
    for node in network.nodes.values():
        if node.virtual_incoming_link is not None:
            vin.append(node.virtual_incoming_link)
            vout.append(node.virtual_outgoing_link)
    for tag, vl in (("vin", vin), ("vout", vout)):
        for name in VLINK_F64:
            if vl:
                out[f"{tag}_{name}"] = np.stack([np.asarray(getattr(l, name), dtype=np.float64)[sl] for l in vl])
    return out


def run_reference(name, steps=None, seed=0, replica=0, mode="philox", mutate=None, np_seed=None,
                  record_tf=False, demand_override=None, create_kwargs=None, network=None):
    """Build `name` with the reference's NetworkEnvGenerator and run network_loading under the
    injected RNG.  `mutate(network, t)` is called after step t (examples/six_node.py:29-30 style).
    Returns (network, static dump, state dump, extras)."""
    ref = load_reference()
    if np_seed is not None:
        np.random.seed(np_seed)
    if network is None:
        gen = ref["env"].NetworkEnvGenerator()
        network = gen.create_network(name, **(create_kwargs or {}))
    if demand_override:
        for nid, arr in demand_override.items():
            network.nodes[nid].demand = np.array(arr, dtype=np.float64)
    T = network.simulation_steps
    last = T if steps is None else min(T, steps)
    static = dump_static(network)
    tf_hist = {}
    with InjectedRNG(network, seed=seed, replica=replica, mode=mode) as inj:
        if mutate is not None:
            mutate(network, 0)          # changes made before the first step (examples/forky_queues.py:112)
        for t in range(1, last):
            network.network_loading(t)
            if record_tf:
                for nid, node in network.nodes.items():
                    tf_hist.setdefault(int(nid), []).append(np.array(node.turning_fractions, dtype=np.float64))
            if mutate is not None:
                mutate(network, t)
    state = dump_state(network, steps=None if steps is None else last)
    extras = {"draws": dict(inj.draws), "steps_run": last}
    if record_tf:
        extras["tf_hist"] = {k: np.stack(v) for k, v in tf_hist.items()}
    return network, static, state, extras


# ------------------------------------------------------------------------------------------------ RL caller (config #5)
def load_reference_rl():
    """The reference's RL glue.  `rl/__init__.py` imports pettingzoo/gymnasium, which this image lacks, so the package
    initialiser is bypassed: a bare package object stands in for `rl` and the REAL modules rl/discovery.py and
    rl/builders.py (numpy only) are imported under it.  The env class itself (rl/pz_pednet_env.py) cannot be imported;
    its two pure functions `_compute_rewards` (:548-581) and `_check_terminations` (:583-624) are compiled from the
    reference's own source text, unmodified, and called on a small stand-in carrying the attributes they read."""
    if "rl_mods" in _loaded:
        return _loaded["rl_mods"]
    load_reference()
    import ast
    import importlib
    import types

    if "rl" not in sys.modules:
        pkg = types.ModuleType("rl")
        pkg.__path__ = [os.path.join(REF_ROOT, "rl")]
        sys.modules["rl"] = pkg
    discovery = importlib.import_module("rl.discovery")
    builders = importlib.import_module("rl.builders")
    path = os.path.join(REF_ROOT, "rl", "pz_pednet_env.py")
    tree = ast.parse(open(path).read(), filename=path)
    wanted = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.ClassDef) and node.name == "PedNetParallelEnv":
            for item in node.body:
                if isinstance(item, ast.FunctionDef) and item.name in ("_compute_rewards", "_check_terminations"):
                    mod = ast.Module(body=[item], type_ignores=[])
                    ns = {"np": np, "Dict": dict}
                    exec(compile(mod, path, "exec"), ns)
                    wanted[item.name] = ns[item.name]
    _loaded["rl_mods"] = {"discovery": discovery, "builders": builders, **wanted}
    return _loaded["rl_mods"]


class RefEnvShim:
    """State the extracted `_compute_rewards` / `_check_terminations` read (pz_pednet_env.py:49-116)."""

    def __init__(self, network, obs_mode="option3", normalize_obs=False, action_gap=1):
        rl = load_reference_rl()
        self.network = network
        self.sim_step = 1
        self.simulation_steps = network.params["simulation_steps"]
        self.agent_manager = rl["discovery"].AgentManager(network)
        self.possible_agents = self.agent_manager.get_all_agent_ids()
        self.obs_builder = rl["builders"].ObservationBuilder(network, self.agent_manager, normalize_obs, obs_mode)
        ut = network.params["unit_time"]
        self.action_applier = rl["builders"].ActionApplier(network, self.agent_manager, 0.25 * ut, 0.25 * ut, 1.5)
        self._action_gap = action_gap
        self._rl = rl

    def step(self, actions):
        """pz_pednet_env.py:195-254 restated around the real components."""
        if len(actions) > 0:
            self.action_applier.apply_all_actions(actions)
        cumulative = {a: 0.0 for a in self.possible_agents}
        for _ in range(self._action_gap):
            self.network.network_loading(self.sim_step)
            obs = {a: self.obs_builder.build_observation(a, self.sim_step) for a in self.possible_agents}
            for a, r in self._rl["_compute_rewards"](self).items():
                cumulative[a] += r
            term = self._rl["_check_terminations"](self)
            self.sim_step += 1
        return obs, cumulative, term
