"""Per-replica scenarios on one topology (device-side reset / scenario randomisation).

The reference randomises a scenario by rebuilding the whole ``Network`` with overrides
(``NetworkEnvGenerator.randomize_network`` -> ``create_network(name, od_flows=..., link_params_overrides=...,
demand_params_overrides=...)``, src/utils/env_loader.py:160-181, 81-158).  Link parameters, OD weights and demand do
not change the topology or the routes (paths are searched on link lengths), so on the GPU every replica keeps the shared
CSR tables and only carries its own

  * k_critical / k_jam / free_flow_speed per link (+ the free-flow look-back, shock-wave look-back and travel_time[0]
    derived from them with the reference's own expressions),
  * time-constant OD weights,
  * origin demand arrays,

which ``ScenarioBatch.commit()`` uploads before resetting the state.  ``generate_random_od_nodes`` (:261-359) changes
which nodes own virtual links, i.e. the topology; that part of ``randomize_network`` cannot be expressed per replica and
is not applied here (documented gap).
"""
import copy

import numpy as np

from .od_manager import DemandGenerator


def derive_statics(length, vf, kc, kj, unit_time):
    """(travel_time[0], free_flow_tau, tau_shockwave) with the expressions of link.py:58-63,83-86,380."""
    capacity = vf * kc
    shockwave_speed = capacity / (kj - kc)
    tt0 = np.float32(min(length / vf, length / 0.05))
    return tt0, round(tt0 / unit_time), round(length / (shockwave_speed * unit_time))


def derive_statics_arrays(length, vf, kc, kj, unit_time):
    """``derive_statics`` for arrays (length [L, 1], the others [L, R]) with the same roundings: ``round`` of a float32 /
    float64 quotient is round-half-even = ``np.rint``; ``np.float32 / unit_time`` is a float32 division (NEP 50)."""
    shockwave_speed = (vf * kc) / (kj - kc)
    tt0 = np.minimum(length / vf, length / 0.05).astype(np.float32)
    fft = np.rint(tt0 / np.float32(unit_time)).astype(np.int32)
    tau_sw = np.rint(length / (shockwave_speed * unit_time)).astype(np.int32)
    return tt0, fft, tau_sw


def merged_link_config(base_params, edge_distances, link_params_overrides):
    """``params['links']`` as create_network builds it (env_loader.py:93-144) from a base config and overrides."""
    links = copy.deepcopy(base_params.get("links", {}))
    defaults = base_params["default_link"]
    for link_id, ov in (link_params_overrides or {}).items():
        links.setdefault(link_id, {}).update(ov)
    if edge_distances:
        for (u, v), distance in edge_distances.items():
            key = f"{u}_{v}"
            merged = defaults.copy()
            merged.update(links.get(key, {}))
            merged["length"] = distance
            links[key] = merged
            if f"{v}_{u}" not in links:
                links[f"{v}_{u}"] = merged
    return links


def _pair_of(link_id):
    u, v = link_id.split("_")
    return frozenset((int(u), int(v)))


class ScenarioBatch:
    """Collects one scenario per replica for a ``Network`` and uploads them together."""

    def __init__(self, network, edge_distances=None):
        self.net = network
        self.edge_distances = edge_distances
        L, R = network.n_links, network.n_replicas
        links = network._link_list
        base = {"kc": ("k_critical", np.float64), "kj": ("k_jam", np.float64), "vf": ("free_flow_speed", np.float64),
                "fft": ("free_flow_tau", np.int32), "tau_sw": ("tau_shockwave", np.int32), "tt0": ("travel_time0", np.float32)}
        self._base = {k: np.array([getattr(l, attr) for l in links], dtype=dt) for k, (attr, dt) in base.items()}
        # the matrices [L, R] (attributes kc, kj, vf, fft, tau_sw, tt0, od_w): host arrays -- or, after draw_random, whatever the device
        # drew, fetched when somebody looks (see _matrix)
        self._arr = {k: np.repeat(col[:, None], R, axis=1) for k, col in self._base.items()}
        self._on_device = set()
        # an override of corridor "u_v" only changes the links between u and v (create_network merges link by link,
        # env_loader.py:93-144): everything below works on the touched node pairs instead of the whole link table
        self._pair_links, self._pair_edges = {}, {}
        for l in links:
            self._pair_links.setdefault(frozenset((l.start_node.node_id, l.end_node.node_id)), []).append(l)
        for (u, v), d in (edge_distances or {}).items():        # dict order matters inside a pair
            self._pair_edges.setdefault(frozenset((int(u), int(v))), []).append(((u, v), d))
        self._arr["od_w"] = None
        if network.od_manager is not None:
            w0 = network.od_manager.as_matrix()
            self._arr["od_w"] = np.repeat(w0[:, :1], R, axis=1)
            self._od_index = {od: i for i, od in enumerate(network.od_manager.od_flows.keys())}
        self.demand = {}         # (node_id, replica) -> array
        self.link_params_dirty = self.od_dirty = False

    def _matrix(self, name):
        if name in self._on_device:          # drawn by pedn_randomize_scenarios: the host copy is made on demand
            eng = self.net._flush()
            if name == "od_w":
                self._arr["od_w"] = eng.get_od_weights_per_replica()
                self._on_device.discard("od_w")
            else:
                self._arr.update(eng.get_link_params())
                self._on_device -= {"kc", "kj", "vf", "fft", "tau_sw", "tt0"}
        return self._arr[name]

    def _link_overrides(self, r, overrides):
        net = self.net
        A = {k: self._matrix(k) for k in self._base}     # the [L, R] matrices once per call (they are properties: fetched lazily)
        for k, col in self._base.items():        # the scenario of replica r starts from the base configuration
            A[k][:, r] = col
        pairs = {_pair_of(link_id) for link_id in overrides}
        edges = {}
        for pr in pairs:
            edges.update(dict(self._pair_edges.get(pr, [])))
        sub = dict(net.params)
        keys = {f"{a}_{b}" for pr in pairs for a in pr for b in pr if a != b}
        sub["links"] = {k: v for k, v in net.params.get("links", {}).items() if k in keys}
        cfg = merged_link_config(sub, edges if self.edge_distances else None, overrides)
        default = net.params.get("default_link", {})
        for pr in pairs:
            for link in self._pair_links.get(pr, []):
                i, j = link.start_node.node_id, link.end_node.node_id
                a, b = (i, j) if i < j else (j, i)
                lp = default
                for key in (f"{a}_{b}", f"{b}_{a}"):         # network.py:169-192: forward key, then reverse key
                    if key in cfg:
                        lp = {**default, **cfg[key]}
                        break
                for name in ("length", "width"):
                    if lp[name] != getattr(link, "length" if name == "length" else "_width"):
                        raise ValueError(f"per-replica override of '{name}' is not supported (link {link.link_id})")
                vf, kc, kj = lp["free_flow_speed"], lp["k_critical"], lp["k_jam"]
                tt0, fft, tsw = derive_statics(link.length, vf, kc, kj, net.unit_time)
                k = link.index
                A["kc"][k, r], A["kj"][k, r], A["vf"][k, r] = kc, kj, vf
                A["fft"][k, r], A["tau_sw"][k, r], A["tt0"][k, r] = fft, tsw, tt0

    def set_replica(self, r, link_params_overrides=None, od_flows=None, demand=None, demand_params_overrides=None):
        """Scenario of replica ``r`` in the vocabulary of ``create_network``.  ``demand`` maps node id -> array and wins over
        ``demand_params_overrides`` (which regenerates the origin demand with numpy's global RNG like the reference)."""
        net = self.net
        if link_params_overrides is not None:
            self._link_overrides(r, link_params_overrides)
            self.link_params_dirty = True
        if od_flows is not None:
            if self.od_w is None:
                raise ValueError("the scenario has no destination nodes / OD weights")
            for od, w in od_flows.items():
                arr = np.asarray(w, dtype=np.float64).reshape(-1)
                if arr.size > 1 and not np.all(arr == arr[0]):
                    raise ValueError("per-replica OD weights must be constant in time")
                self.od_w[self._od_index[tuple(od)], r] = arr[0]
            self.od_dirty = True
        if demand_params_overrides is not None:
            params = dict(net.params)
            params["demand"] = {k: dict(v) for k, v in net.params.get("demand", {}).items()}
            for key, ov in demand_params_overrides.items():
                params["demand"].setdefault(key, {}).update(ov)
            gen = DemandGenerator(net.simulation_steps, params, None)
            for node in net.nodes.values():           # node creation order, like Network.__init__
                if node.virtual_incoming_link is not None and node.node_id in net.origin_nodes:
                    oc = params.get("demand", {}).get(f"origin_{node.node_id}", {})
                    self.demand[(node.node_id, r)] = np.asarray(gen.generate_custom(node.node_id, oc.get("pattern", "gaussian_peaks")), dtype=np.float64)
        for nid, arr in (demand or {}).items():
            self.demand[(nid, r)] = np.asarray(arr, dtype=np.float64)

    def draw_random(self, seed, link_fraction=0.2):
        """A new scenario for EVERY replica drawn ON THE DEVICE, in place (``pedn_randomize_scenarios``): the distributions of the
        reference's randomisers -- see ``draw_random_host``, the same thing in numpy -- keyed (seed, global replica id) with
        Philox, so a function of the seed that does not depend on how the ensemble is sharded.  Nothing is uploaded; ``kc`` ...
        ``od_w`` are fetched from the device when read.  Call ``commit`` for the reset."""
        net = self.net
        eng = net._flush()
        origins = [node for node in net.nodes.values() if node.virtual_incoming_link is not None and node.node_id in net.origin_nodes]
        # seed None = fresh entropy every time, like np.random.default_rng(None) in draw_random_host and like the reference, which keeps
        # consuming its np.random stream (the usual Gym pattern seeds the first reset only): never the same scenario twice
        self.last_seed = int(np.random.SeedSequence().entropy) & (2 ** 64 - 1) if seed is None else int(seed)
        eng.randomize_scenarios(self.last_seed, link_fraction, links=True, od_weights=self._arr["od_w"] is not None,
                                origin_nodes=[node.index for node in origins])
        if int(len(self._pair_links) * link_fraction) > 0:
            self._on_device |= {"kc", "kj", "vf", "fft", "tau_sw", "tt0"}
        if self._arr["od_w"] is not None:
            self._on_device.add("od_w")
        ids = {node.node_id for node in origins}
        self.demand = {key: v for key, v in self.demand.items() if key[0] not in ids}
        for node in origins:
            net._dirty_demand.discard(node)
        self.link_params_dirty = self.od_dirty = False

    def draw_random_host(self, seed, link_fraction=0.2):
        """``draw_random`` in numpy on the host (six [L, R] uploads; 19-24 ms for 2048 envs where the device version takes < 1 ms):
        kept as the cross-check of the device kernels' distributions.
        A new scenario for EVERY replica, drawn for all replicas at once from the distributions of the reference's
        randomisers: ``generate_random_link_params`` (env_loader.py:363-424: ``link_fraction`` of the corridors; with
        probability 1/2 k_critical and k_jam scaled by U(0.6, 1.2) with the floors max(0.5, .) / max(2 k_c, .); with
        probability 1/2 free_flow_speed scaled by U(0.6, 0.9)), ``generate_random_od_flows`` (:224-259: U(1, 10) per OD pair)
        and ``generate_random_demand_params`` (:183-222: pattern, base U(2, 10), peak max(U(10, 30), base + 5)) with the demand
        series of od_manager.py:92-112 drawn on the device (``pedn_draw_demand``).  One numpy Generator replaces the
        reference's np.random stream per env, so this is the same distribution, not the same numbers -- for those use
        ``set_replica`` with the mirrored ``generate_random_*``.  A perturbed corridor scales each of its links from that
        link's own base parameters.  Uploads immediately; call ``commit`` for the reset."""
        net = self.net
        rng = np.random.default_rng(seed)
        L, R = net.n_links, net.n_replicas
        pairs = list(self._pair_links.keys())
        P = len(pairs)
        k = int(P * link_fraction)
        if k > 0:
            chosen = np.zeros((R, P), dtype=bool)
            pick = np.argsort(rng.random((R, P)), axis=1)[:, :k]          # k corridors per replica without replacement
            np.put_along_axis(chosen, pick, True, axis=1)
            dens = chosen & (rng.random((R, P)) < 0.5)
            f = rng.uniform(0.6, 1.2, (R, P))
            spd = chosen & (rng.random((R, P)) < 0.5)
            g = rng.uniform(0.6, 0.9, (R, P))
            pair_of_link = np.empty(L, dtype=np.int64)
            for p, pr in enumerate(pairs):
                for l in self._pair_links[pr]:
                    pair_of_link[l.index] = p
            D, F = dens[:, pair_of_link].T, f[:, pair_of_link].T              # [L, R]
            S, G = spd[:, pair_of_link].T, g[:, pair_of_link].T
            kc0, kj0, vf0 = (self._base[n][:, None] for n in ("kc", "kj", "vf"))
            self.kc = np.where(D, np.maximum(0.5, kc0 * F), kc0)
            self.kj = np.where(D, np.maximum(self.kc * 2.0, kj0 * F), kj0)
            self.vf = np.where(S, vf0 * G, vf0)
            length = np.array([l.length for l in net._link_list], dtype=np.float64)[:, None]
            self.tt0, self.fft, self.tau_sw = derive_statics_arrays(length, self.vf, self.kc, self.kj, net.unit_time)
            self.link_params_dirty = True
        if self.od_w is not None:
            self.od_w = rng.uniform(1.0, 10.0, self.od_w.shape)
            self.od_dirty = True
        eng = net._flush()
        T = net.simulation_steps
        for node in net.nodes.values():
            if node.virtual_incoming_link is None or node.node_id not in net.origin_nodes:
                continue
            pattern = rng.integers(0, 3, R)                                   # gaussian_peaks / constant / sudden_demand
            base = rng.uniform(2.0, 10.0, R)
            peak = np.maximum(rng.uniform(10.0, 30.0, R), base + 5.0)
            spike_len = rng.integers(10, 20, R)
            spike_start = rng.integers(0, np.maximum(1, T - spike_len))
            spike_height = rng.integers(20, 50, R).astype(np.float64)
            eng.draw_demand(node.index, int(rng.integers(0, 2 ** 63)), pattern, base, peak, spike_start, spike_len, spike_height)
            self.demand = {key: v for key, v in self.demand.items() if key[0] != node.node_id}

    def commit(self, reset=True):
        """Upload everything that changed and (by default) reset the state: travel_time[0] depends on the parameters.

        A demand array REPLACES the replica's whole row: one shorter than T + 1 is followed by zeros (like ``pedn_set_demand``
        and like the reference, where ``node.demand`` is the array the generator returned and nothing else), whether the batch
        covers every replica (one matrix upload) or a subset (``pedn_set_demand_rows``); replicas without an entry keep theirs."""
        net = self.net
        eng = net._flush()
        if self.link_params_dirty:
            eng.set_link_params(self.kc, self.kj, self.vf, self.fft, self.tau_sw, self.tt0)
        if self.od_dirty:
            eng.set_od_weights_per_replica(self.od_w)
        by_node = {}
        for (nid, r), arr in self.demand.items():
            by_node.setdefault(nid, {})[r] = arr
        R, T1 = net.n_replicas, net.simulation_steps + 1
        for nid, rows in by_node.items():                   # one upload per origin, whatever subset of replicas it covers
            reps = sorted(rows)
            mat = np.zeros((len(reps), T1))
            for k, r in enumerate(reps):
                n = min(len(rows[r]), T1)
                mat[k, :n] = rows[r][:n]
            if len(reps) == R:
                eng.set_demand_matrix(net.nodes[nid].index, mat)
            else:
                eng.set_demand_rows(net.nodes[nid].index, reps, mat)
            net._dirty_demand.discard(net.nodes[nid])
        self.link_params_dirty = self.od_dirty = False
        self.demand = {}
        net._invalidate()
        if reset:
            net.reset()


def _matrix_property(name):
    def get(self):
        return self._matrix(name)

    def put(self, value):
        self._on_device.discard(name)
        self._arr[name] = value

    return property(get, put)


for _name in ("kc", "kj", "vf", "fft", "tau_sw", "tt0", "od_w"):
    setattr(ScenarioBatch, _name, _matrix_property(_name))
