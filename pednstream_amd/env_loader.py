"""``NetworkEnvGenerator`` -- scenario directory -> ``Network`` (drop-in boundary).

Interface mirrored from /root/reference/src/utils/env_loader.py: ``NetworkEnvGenerator(data_dir="data")`` :24,
``load_network_data`` :34-79, ``create_network(name, custom_demand_functions, od_flows, link_params_overrides,
demand_params_overrides)`` :81-158, ``randomize_network(name, seed, randomize_params)`` :160-181 and the
``generate_random_*`` scenario randomisers :183-424, including the reference's behaviour of caching and
mutating ``self.config`` in place across calls.  Extra keyword arguments (``verbose``, ``n_replicas``, ...) are
forwarded to ``Network``; ``verbose`` is accepted because rl/pz_pednet_env.py:80,168 passes it.

A scenario directory may hold the reference's native files (``sim_params.yaml``, ``adj_matrix.npy``,
``edge_distances.pkl``, ``node_positions.json``) or the neutral pair written by tools/import_scenarios.py
(``scenario.json`` + ``topology.npz``).
"""
import json
import os
import pickle
from pathlib import Path
from typing import Callable, List

import numpy as np

from .config import load_config
from .network import Network


class NetworkEnvGenerator:
    def __init__(self, data_dir="data"):
        p = Path(data_dir)
        if not p.is_absolute():
            here = Path.cwd() / p
            p = here if here.is_dir() else Path(__file__).resolve().parent.parent / data_dir
        self.data_dir = p
        self.network = None
        self.network_data = None
        self.config = None

    # ---------------------------------------------------------------------------------------------- loading
    def load_network_data(self, data_path: str) -> dict:
        folder = os.path.join(self.data_dir, f"{data_path}")
        yaml_path = os.path.join(folder, "sim_params.yaml")
        json_path = os.path.join(folder, "scenario.json")
        if os.path.exists(yaml_path):
            self.config = load_config(yaml_path)
        elif os.path.exists(json_path):
            self.config = load_config(json_path)
        else:
            raise FileNotFoundError(f"Network data file not found: {yaml_path}")

        edge_distances = adjacency = positions = None
        topo_path = os.path.join(folder, "topology.npz")
        topo = np.load(topo_path) if os.path.exists(topo_path) else None

        if os.path.exists(os.path.join(folder, "edge_distances.pkl")):
            with open(os.path.join(folder, "edge_distances.pkl"), "rb") as f:
                edge_distances = pickle.load(f)
        elif topo is not None and "edge_uv" in topo:
            edge_distances = {(int(u), int(v)): float(d) for (u, v), d in zip(topo["edge_uv"], topo["edge_dist"])}

        if "adjacency_matrix" in self.config:
            adjacency = self.config["adjacency_matrix"]
        elif os.path.exists(os.path.join(folder, "adj_matrix.npy")):
            adjacency = np.load(os.path.join(folder, "adj_matrix.npy"))
        elif topo is not None and "adj_rows" in topo:
            n = int(topo["n_nodes"])
            adjacency = np.zeros((n, n), dtype=np.int64)
            adjacency[topo["adj_rows"], topo["adj_cols"]] = 1
        else:
            raise FileNotFoundError(f"no adjacency matrix for scenario {data_path}")

        if os.path.exists(os.path.join(folder, "node_positions.json")):
            with open(os.path.join(folder, "node_positions.json"), "r") as f:
                positions = {str(k): v for k, v in json.load(f).items()}
        elif topo is not None and "pos_ids" in topo:
            positions = {str(int(k)): [float(x), float(y)] for k, (x, y) in zip(topo["pos_ids"], topo["pos_xy"])}

        return {"adjacency_matrix": adjacency, "edge_distances": edge_distances, "node_positions": positions}

    def create_network(self, yaml_file_path: str, custom_demand_functions: List[Callable] = None,
                       od_flows: dict = None, link_params_overrides: dict = None,
                       demand_params_overrides: dict = None, **network_kwargs):
        if self.network_data is None:
            self.network_data = self.load_network_data(yaml_file_path)
        params = self.config["params"]
        defaults = params["default_link"]

        if link_params_overrides:
            params.setdefault("links", {})
            for link_id, ov in link_params_overrides.items():
                params["links"].setdefault(link_id, {}).update(ov)
        if od_flows:
            self.config["od_flows"] = od_flows
        if demand_params_overrides:
            params.setdefault("demand", {})
            for origin_key, ov in demand_params_overrides.items():
                params["demand"].setdefault(origin_key, {}).update(ov)
        params.setdefault("links", {})

        if self.network_data["edge_distances"]:
            for (u, v), distance in self.network_data["edge_distances"].items():
                key = f"{u}_{v}"
                merged = defaults.copy()
                merged.update(params["links"].get(key, {}))
                merged["length"] = distance
                params["links"][key] = merged
                if f"{v}_{u}" not in params["links"]:
                    params["links"][f"{v}_{u}"] = merged

        self.network = Network(
            adjacency_matrix=self.network_data["adjacency_matrix"], params=params,
            origin_nodes=self.config.get("origin_nodes", []),
            destination_nodes=self.config.get("destination_nodes", []),
            demand_pattern=custom_demand_functions, od_flows=self.config.get("od_flows", None),
            pos=self.network_data.get("node_positions"), **network_kwargs)
        # optional tuning artefact of this repository (not a reference file): the measured cost of every node's slowest node-kernel wave,
        # by which the engine packs nodes into workgroups (tools/pack_calibrate.py; include/pedn.h: pedn_model_desc.node_cost)
        cost_path = os.path.join(self.data_dir, f"{yaml_file_path}", "pack_cost.json")
        if os.path.exists(cost_path):
            with open(cost_path) as f:
                self.network.node_pack_cost = {int(k): float(v) for k, v in json.load(f)["node_cost"].items()}
        return self.network

    # ---------------------------------------------------------------------------------------------- randomisers
    def randomize_network(self, yaml_file_path: str, seed: int = None, randomize_params: dict = None,
                          **network_kwargs):
        self.generate_random_od_nodes(seed)
        link_ov = self.generate_random_link_params(seed)
        od_w = self.generate_random_od_flows(seed)
        demand_ov = self.generate_random_demand_params(seed)
        return self.create_network(yaml_file_path, od_flows=od_w, link_params_overrides=link_ov,
                                   demand_params_overrides=demand_ov, **network_kwargs)

    def generate_random_demand_params(self, seed: int = None) -> dict:
        if seed is not None:
            np.random.seed(seed)
        out = {}
        for origin in self.config.get("origin_nodes", []):
            pattern = np.random.choice(["gaussian_peaks", "constant", "sudden_demand"])
            base = np.random.uniform(2.0, 10.0)
            peak = np.random.uniform(10.0, 30.0)
            if peak < base + 5:
                peak = base + 5
            out[f"origin_{origin}"] = {"pattern": pattern, "base_lambda": float(base), "peak_lambda": float(peak),
                                       "seed": seed}
        return out

    def generate_random_od_flows(self, seed: int = None) -> dict:
        if seed is not None:
            np.random.seed(seed)
        steps = self.config["params"]["simulation_steps"]
        out = {}
        for o in self.config.get("origin_nodes", []):
            for d in self.config.get("destination_nodes", []):
                if o == d:
                    continue
                out[(o, d)] = np.full(steps + 1, np.random.uniform(1.0, 10.0))
        return out

    def generate_random_od_nodes(self, seed: int = None) -> dict:
        if seed is not None:
            np.random.seed(seed)
        adj = self.network_data["adjacency_matrix"]
        controllers = self.network.controller_nodes

        def hop2(seed_nodes):
            first = set()
            for n in seed_nodes:
                first.update(np.where(adj[n, :] == 1)[0].tolist())
            second = set()
            for n in first:
                second.update(np.where(adj[n, :] == 1)[0].tolist())
            first.update(second)
            return list(first)

        origins = self.config.get("origin_nodes", []).copy()
        if np.random.random() < 0.5:
            cand = [n for n in hop2(origins) if n not in origins and n not in controllers]
            if cand:
                k = np.random.randint(1, min(2, len(cand) + 1))
                origins.extend(int(x) for x in np.random.choice(cand, k, replace=False))
        if len(origins) > 1 and np.random.random() < 0.5:
            k = np.random.randint(1, min(2, len(origins)))
            drop = np.random.choice(len(origins), k, replace=False)
            origins = [o for i, o in enumerate(origins) if i not in drop]
        if np.random.random() < 0.5:
            victim = np.random.choice(origins)
            cand = [n for n in hop2([victim]) if n not in origins and n not in controllers]
            if cand:
                origins[origins.index(victim)] = int(np.random.choice(cand))

        dests = self.config.get("destination_nodes", []).copy()
        if np.random.random() < 0.5:
            cand = [n for n in hop2(dests) if n not in dests and n not in controllers]
            if cand:
                k = np.random.randint(1, min(3, len(cand) + 1))
                dests.extend(int(x) for x in np.random.choice(cand, k, replace=False))
        if len(dests) > len(origins) and np.random.random() < 0.5:
            removable = [d for d in dests if d not in origins]
            if removable:
                k = np.random.randint(1, min(2, len(removable) + 1))
                gone = [int(x) for x in np.random.choice(removable, k, replace=False)]
                dests = [d for d in dests if d not in gone]

        origins = [int(x) for x in origins]
        dests = [int(x) for x in dests]
        self.config["origin_nodes"] = origins
        self.config["destination_nodes"] = dests
        return {"origin_nodes": origins, "destination_nodes": dests}

    def generate_random_link_params(self, seed: int = None) -> dict:
        if seed is not None:
            np.random.seed(seed)
        # the corridor list depends on the scenario's topology only: built once per loaded network (the batched env calls this once
        # per env and reset).  np.random.choice(list, k, replace=False) draws permutation(len(list))[:k] whatever the elements are, so
        # choosing INDICES consumes the same random numbers as the reference's choice over the strings
        cache = getattr(self, "_corridor_cache", None)
        if cache is None or cache[0] is not self.network_data:
            corridors = []
            ed = self.network_data.get("edge_distances") if self.network_data else None
            if ed:
                corridors = [f"{u}_{v}" for (u, v) in ed.keys() if u < v]
            elif self.network_data and "adjacency_matrix" in self.network_data:
                rows, cols = np.where(self.network_data["adjacency_matrix"] == 1)
                corridors = [f"{u}_{v}" for u, v in zip(rows, cols) if u < v]
            cache = self._corridor_cache = (self.network_data, corridors)
        corridors = cache[1]
        params = self.config["params"]
        defaults = params["default_link"]
        out = {}
        if corridors:
            k = int(len(corridors) * 0.2)
            if k > 0:
                for idx in np.random.choice(len(corridors), k, replace=False):
                    link_id = corridors[idx]
                    cur = params["links"].get(link_id, {})
                    ov = {}
                    if np.random.random() < 0.5:
                        f = np.random.uniform(0.6, 1.2)
                        ov["k_critical"] = max(0.5, cur.get("k_critical", defaults["k_critical"]) * f)
                        ov["k_jam"] = max(ov["k_critical"] * 2.0, cur.get("k_jam", defaults["k_jam"]) * f)
                    if np.random.random() < 0.5:
                        ov["free_flow_speed"] = cur.get("free_flow_speed", defaults["free_flow_speed"]) * np.random.uniform(0.6, 0.9)
                    if ov:
                        out[link_id] = ov
        return out
