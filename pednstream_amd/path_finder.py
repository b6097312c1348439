"""Route enumeration and per-node turn tables (host side, setup time only).

The per-step logit route choice (reference /root/reference/src/LTM/path_finder.py:561-737) runs on the GPU;
what stays on the host is the one-off construction of its static tables, which this module reproduces
from the reference's behaviour (path_finder.py:146-268, 304-546):

  * k shortest simple paths per OD pair (networkx ``shortest_simple_paths`` on a DiGraph whose edges are
    inserted in ``network.links`` order -- insertion order is part of the tie-breaking),
  * detour expansion at controller nodes ('penalize' mode, factor 2, <= 3 detours per neighbour),
  * duplicate-path removal,
  * per node: ``turns_distances[od][up][down]``, ``up_od_probs[up][od]``, ``ods_in_turns[(up, down)]``.

Python ``set``/``dict`` iteration orders of these tables fix the floating-point summation order of the
turning-fraction kernel, so the same container types and insertion sequences as the reference are used and
the orders are then frozen into CSR arrays by ``pednstream_amd.flatten``.
"""
from collections import defaultdict

import numpy as np

VIRTUAL = -1  # node id standing for the virtual (origin/destination) side of a turn


def _first_k_simple_paths(graph, origin, dest, k=None):
    import networkx as nx

    try:
        it = nx.shortest_simple_paths(graph, origin, dest, weight="weight")
    except Exception:
        return []
    found = []
    for p in it:
        found.append(p)
        if k is not None and len(found) >= k:
            break
    return found


class NodeTurnTable:
    """Static route-choice tables of one intersection node (attached to the node view)."""

    def __init__(self):
        self.turns_distances = {}                                    # {od: {up: {down: remaining distance}}}
        self.up_od_probs = defaultdict(lambda: defaultdict(int))     # {up: {od: P(od|up)}}
        self.ods_in_turns = {}                                       # {(up, down): set(od)}
        self.node_turn_probs = {}                                    # {od: {(up, down): P(down|up,od)}} (host mirror, unused by the GPU)


class PathFinder:
    def __init__(self, links, params=None, controller_nodes=None, controller_links=None, logger=None):
        import networkx as nx

        self.links = links
        self.logger = logger
        self.od_paths = {}
        self.nodes_in_paths = set()
        self.node_to_od_pairs = {}
        self.tables = {}            # node_id -> NodeTurnTable
        self._initialized = False

        self.graph = nx.DiGraph()
        for (u, v), link in links.items():
            self.graph.add_edge(u, v, weight=link.length, num_pedestrians=0.0)

        pp = params.get("path_finder", {}) if params else {}
        self.temp = pp.get("temp", 0.1)
        self.alpha = pp.get("alpha", 1.0)
        self.beta = pp.get("beta", 0.05)
        self.omega = pp.get("omega", 0.05)
        self.std_dev = pp.get("std_dev", 0)
        self.epsilon = np.random.normal(0, self.std_dev)     # drawn once, like path_finder.py:163
        self.k_paths = pp.get("k_paths", 3)
        self.verbose = pp.get("verbose", True)

        self.controller_nodes = controller_nodes
        self.controller_links = controller_links
        self.controllers_enabled = bool(controller_nodes or controller_links)
        self.detour_exploration_mode = "penalize"
        self.detour_penalty_factor = 2
        self.max_detour_paths = 3

    # ------------------------------------------------------------------ helpers
    def is_controller_node(self, node_id):
        return bool(self.controllers_enabled and node_id in self.controller_nodes)

    def _log(self, msg):
        if self.logger and self.verbose:
            self.logger.info(msg)

    def _register_path(self, path, od):
        for n in path:
            self.nodes_in_paths.add(n)
            if n not in self.node_to_od_pairs:
                self.node_to_od_pairs[n] = set()
            self.node_to_od_pairs[n].add(od)

    def calculate_path_distance(self, path, start_idx=0):
        dist = 0
        for i in range(start_idx, len(path) - 1):
            edge = self.graph.edges[(path[i], path[i + 1])]
            if edge:
                dist += edge["weight"]
        return dist

    # ------------------------------------------------------------------ path enumeration
    def find_od_paths(self, od_pairs, nodes):
        import networkx as nx

        for origin, dest in od_pairs:
            try:
                paths = _first_k_simple_paths(self.graph, origin, dest, k=self.k_paths)
                self.od_paths[(origin, dest)] = paths
                for p in paths:
                    self._register_path(p, (origin, dest))
            except nx.NetworkXNoPath:
                self._log(f"No path found between {origin} and {dest}")
                self.od_paths[(origin, dest)] = []

        if not self._initialized and self.controllers_enabled:
            for cn in self.controller_nodes:
                for od in self.node_to_od_pairs[cn]:     # KeyError if a controller lies on no OD path (reference quirk, kept)
                    before = len(self.od_paths[od])
                    self.expand_controller_paths(nodes[cn], od)
                    self._log(f"Controller node {cn}: Added {len(self.od_paths[od]) - before} detour path(s) for OD {od}")
        self._drop_duplicate_paths()
        self._build_turn_tables(nodes)

    def _drop_duplicate_paths(self):
        def norm(n):
            try:
                return int(n)
            except Exception:
                return str(n)

        for od, paths in self.od_paths.items():
            as_tuples = [tuple(norm(n) for n in p) for p in (paths or [])]
            uniq = set(as_tuples)
            if len(uniq) != len(as_tuples):
                self._log(f"Warning: duplicate paths detected for OD {od}: {len(as_tuples) - len(uniq)} duplicate(s)")
                self.od_paths[od] = [list(p) for p in uniq]

    def expand_controller_paths(self, node, od_pair):
        """Add detours through off-path neighbours of a controller node (reference :304-458)."""
        import networkx as nx

        here = node.node_id
        origin, dest = od_pair
        paths = self.od_paths[od_pair]
        added = []

        out_neighbours = set()
        for link in node.outgoing_links:
            if link.end_node is not None:
                out_neighbours.add(link.end_node.node_id)

        # penalise every edge already used by this OD, more strongly the farther it is from the destination
        penalised = self.graph.copy()
        used = {}
        for p in paths:
            for a, b in zip(p[:-1], p[1:]):
                if (a, b) not in used:
                    try:
                        used[(a, b)] = nx.shortest_path_length(self.graph, b, dest, weight="weight")
                    except nx.NetworkXNoPath:
                        used[(a, b)] = 0
        if self.detour_exploration_mode == "remove":
            penalised.remove_edges_from([e for e in used if penalised.has_edge(*e)])
        elif used:
            far = max(used.values()) if used.values() else 1
            for (a, b), d2d in used.items():
                if penalised.has_edge(a, b):
                    if far > 0:
                        factor = 1.0 + (self.detour_penalty_factor - 1.0) * (d2d / far)
                    else:
                        factor = self.detour_penalty_factor
                    penalised[a][b]["weight"] = penalised[a][b].get("weight", 1) * factor

        for path in paths:
            try:
                at = path.index(here)
            except ValueError:
                continue
            if here == dest:
                continue
            if here == origin:
                up = VIRTUAL
            else:
                up = path[at - 1] if at > 0 else VIRTUAL
            on_path_down = path[at + 1] if at < len(path) - 1 else None
            for nb in out_neighbours:
                if nb == on_path_down or nb == up:
                    continue
                if nb in set(path[:at]):
                    continue
                try:
                    detours = _first_k_simple_paths(penalised, nb, dest, k=self.max_detour_paths)
                    if not detours:
                        continue
                    seen = set(path[:at + 1])
                    for tail in detours:
                        if set(tail[1:]) & seen:
                            continue
                        candidate = path[:at + 1] + tail
                        if tuple(candidate) not in set(tuple(p) for p in self.od_paths[od_pair]):
                            added.append(candidate)
                except Exception:
                    continue

        if added:
            self.od_paths[od_pair].extend(added)
            for p in added:
                self._register_path(p, od_pair)
        return added

    # ------------------------------------------------------------------ turn tables
    def _build_turn_tables(self, nodes):
        for nid in self.nodes_in_paths:
            if nodes[nid].source_num > 2:
                self._build_node_table(nodes[nid])
        self._initialized = True

    def _build_node_table(self, node):
        here = node.node_id
        for od in self.node_to_od_pairs.get(here, set()):
            origin, dest = od
            best = {}
            turn = None
            for path in self.od_paths[od]:
                try:
                    at = path.index(here)
                except ValueError:
                    continue
                if here == origin:
                    turn = (VIRTUAL, path[at + 1])
                elif here == dest:
                    turn = (path[at - 1], VIRTUAL)
                elif at < len(path) - 1:
                    turn = (path[at - 1], path[at + 1])
                remaining = self.calculate_path_distance(path, start_idx=at)
                if turn not in best or remaining < best[turn]:
                    best[turn] = remaining
                    if not self._initialized:
                        tbl = self.tables.setdefault(here, NodeTurnTable())
                        if turn not in tbl.ods_in_turns:
                            tbl.ods_in_turns[turn] = set()
                        tbl.ods_in_turns[turn].add(od)
            if best:
                tbl = self.tables.setdefault(here, NodeTurnTable())
                tbl.turns_distances[od] = {}
                for (up, down), dist in best.items():
                    tbl.turns_distances[od].setdefault(up, {})[down] = dist
                    tbl.up_od_probs[up][od] = 0
                if od not in tbl.node_turn_probs:
                    tbl.node_turn_probs[od] = {}
        tbl = self.tables.get(here)
        if tbl is not None:
            # expose the tables on the node view under the reference's attribute names
            node.turns_distances = tbl.turns_distances
            node.up_od_probs = tbl.up_od_probs
            node.ods_in_turns = tbl.ods_in_turns
            node.node_turn_probs = tbl.node_turn_probs
