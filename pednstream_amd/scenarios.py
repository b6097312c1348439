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


class ScenarioBatch:
    """Collects one scenario per replica for a ``Network`` and uploads them together."""

    def __init__(self, network, edge_distances=None):
        self.net = network
        self.edge_distances = edge_distances
        L, R = network.n_links, network.n_replicas
        links = network._link_list
        self.kc = np.repeat(np.array([[l.k_critical] for l in links], dtype=np.float64), R, axis=1)
        self.kj = np.repeat(np.array([[l.k_jam] for l in links], dtype=np.float64), R, axis=1)
        self.vf = np.repeat(np.array([[l.free_flow_speed] for l in links], dtype=np.float64), R, axis=1)
        self.fft = np.repeat(np.array([[l.free_flow_tau] for l in links], dtype=np.int32), R, axis=1)
        self.tau_sw = np.repeat(np.array([[l.tau_shockwave] for l in links], dtype=np.int32), R, axis=1)
        self.tt0 = np.repeat(np.array([[l.travel_time0] for l in links], dtype=np.float32), R, axis=1)
        self.od_w = None
        if network.od_manager is not None:
            base = network.od_manager.as_matrix()
            self.od_w = np.repeat(base[:, :1], R, axis=1)
            self._od_index = {od: i for i, od in enumerate(network.od_manager.od_flows.keys())}
        self.demand = {}         # (node_id, replica) -> array
        self.link_params_dirty = self.od_dirty = False

    def set_replica(self, r, link_params_overrides=None, od_flows=None, demand=None, demand_params_overrides=None):
        """Scenario of replica ``r`` in the vocabulary of ``create_network``.  ``demand`` maps node id -> array and wins over
        ``demand_params_overrides`` (which regenerates the origin demand with numpy's global RNG like the reference)."""
        net = self.net
        if link_params_overrides is not None:
            cfg = merged_link_config(net.params, self.edge_distances, link_params_overrides)
            default = net.params.get("default_link", {})
            for link in net._link_list:
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
                self.kc[k, r], self.kj[k, r], self.vf[k, r] = kc, kj, vf
                self.fft[k, r], self.tau_sw[k, r], self.tt0[k, r] = fft, tsw, tt0
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
            params = copy.deepcopy(net.params)
            params.setdefault("demand", {})
            for key, ov in demand_params_overrides.items():
                params["demand"].setdefault(key, {}).update(ov)
            gen = DemandGenerator(net.simulation_steps, params, None)
            for node in net.nodes.values():           # node creation order, like Network.__init__
                if node.virtual_incoming_link is not None and node.node_id in net.origin_nodes:
                    oc = params.get("demand", {}).get(f"origin_{node.node_id}", {})
                    self.demand[(node.node_id, r)] = np.asarray(gen.generate_custom(node.node_id, oc.get("pattern", "gaussian_peaks")), dtype=np.float64)
        for nid, arr in (demand or {}).items():
            self.demand[(nid, r)] = np.asarray(arr, dtype=np.float64)

    def commit(self, reset=True):
        """Upload everything that changed and (by default) reset the state: travel_time[0] depends on the parameters."""
        net = self.net
        eng = net._flush()
        if self.link_params_dirty:
            eng.set_link_params(self.kc, self.kj, self.vf, self.fft, self.tau_sw, self.tt0)
        if self.od_dirty:
            eng.set_od_weights_per_replica(self.od_w)
        for (nid, r), arr in self.demand.items():
            eng.set_demand(net.nodes[nid].index, arr, replica=r)
        self.link_params_dirty = self.od_dirty = False
        self.demand = {}
        if reset:
            eng.reset()
            net.current_step = 0
            net._col_cache.clear()
