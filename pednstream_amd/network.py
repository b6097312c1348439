"""``Network`` -- host-side mirror of the reference's network object for the hot path.

Interface mirrored: /root/reference/src/LTM/network.py (``Network.__init__`` :56-121, ``network_loading`` :266-287,
``update_turning_fractions_per_node`` :250-255, attributes ``links/nodes/params/simulation_steps/unit_time/
origin_nodes/destination_nodes/pos/path_finder/od_manager/controller_*``), link surface of
/root/reference/src/LTM/link.py and node surface of /root/reference/src/LTM/node.py.

What is different by design: nodes and links are *views*.  All state lives in HBM as
``[time][link][replica]`` arrays owned by the HIP engine (``pednstream_amd.engine``); ``network_loading(t)``
enqueues the fused node / link kernels for every replica at once, and array attributes such as
``link.density`` fetch their column from the device on access.  ``n_replicas`` independent copies of the
scenario (RL vectorised envs, demand ensembles) are stepped together; the plain reference API addresses
replica 0 on reads and broadcasts on writes, ``network.replica(r)`` scopes both to one replica.
"""
import logging
from typing import Callable, List

import numpy as np

from .od_manager import DemandGenerator, ODManager
from .path_finder import PathFinder

FD_TYPES = {"yperman": 0, "greenshields": 1, "smulders": 2}

# history fields: name -> (field id of the C-ABI, numpy dtype); ids match include/pedn.h
LINK_FIELDS = {
    "inflow": (0, np.float64), "outflow": (1, np.float64),
    "cumulative_inflow": (2, np.float64), "cumulative_outflow": (3, np.float64),
    "sending_flow": (4, np.float64), "receiving_flow": (5, np.float64),
    "back_gate_width_data": (6, np.float64),
    "travel_time": (7, np.float32), "avg_travel_time": (8, np.float32), "num_pedestrians": (9, np.float32),
    "density": (10, np.float32), "speed": (11, np.float32), "link_flow": (12, np.float32),
}
VIRTUAL_FIELDS = ("inflow", "outflow", "cumulative_inflow", "cumulative_outflow")
LINK_DTYPES = {fid: dt for fid, dt in LINK_FIELDS.values()}


class HistoryArray:
    """Lazy view of one ``[T+1]`` history column that lives on the device.

    Indexing with an int fetches one element, anything else materialises the column (cached until the next
    step).  Behaves like a read-only numpy array through ``__array__``."""

    def __init__(self, fetch_col, fetch_one, length, dtype):
        self._fetch_col, self._fetch_one = fetch_col, fetch_one
        self._len, self.dtype = length, np.dtype(dtype)

    def __len__(self):
        return self._len

    @property
    def shape(self):
        return (self._len,)

    def __array__(self, dtype=None, copy=None):
        a = self._fetch_col()
        return a if dtype is None else a.astype(dtype)

    def __getitem__(self, idx):
        if isinstance(idx, (int, np.integer)):
            i = int(idx)
            if i < 0:
                i += self._len
            if not 0 <= i < self._len:
                raise IndexError("index out of range")
            return self._fetch_one(i)
        return self._fetch_col()[idx]

    def __iter__(self):
        return iter(self._fetch_col())

    def tolist(self):
        return self._fetch_col().tolist()

    def copy(self):
        return self._fetch_col().copy()

    def __repr__(self):
        return f"HistoryArray(len={self._len}, dtype={self.dtype})"


class DemandArray(np.ndarray):
    """Origin demand: a real numpy array whose in-place edits are pushed to the device before the next step
    (examples mutate ``network.nodes[n].demand[a:b] = ...``, e.g. examples/long_corridor.py)."""

    def __new__(cls, values, on_write=None):
        obj = np.array(values).view(cls)
        obj._on_write = on_write
        return obj

    def __array_finalize__(self, obj):
        self._on_write = getattr(obj, "_on_write", None)

    def __setitem__(self, key, value):
        super().__setitem__(key, value)
        cb = getattr(self, "_on_write", None)
        if cb is not None:
            cb()


class BaseLinkView:
    """Virtual (origin/destination) link: only flow/cumulative arrays exist (link.py:4-28)."""

    is_virtual = True

    def __init__(self, network, link_id, start_node, end_node, index):
        self._net = network
        self.link_id = link_id
        self.start_node = start_node
        self.end_node = end_node
        self.index = index            # column in the [T+1][L_all][R] flow arrays

    def _hist(self, name):
        fid, dt = LINK_FIELDS[name]
        net, idx = self._net, self.index
        return HistoryArray(lambda: net._read_column(fid, idx), lambda t: net._read_element(fid, idx, t),
                            net.simulation_steps + 1, dt)

    inflow = property(lambda self: self._hist("inflow"))
    outflow = property(lambda self: self._hist("outflow"))
    cumulative_inflow = property(lambda self: self._hist("cumulative_inflow"))
    cumulative_outflow = property(lambda self: self._hist("cumulative_outflow"))
    sending_flow = property(lambda self: -1 * np.ones(self._net.simulation_steps + 1))
    receiving_flow = property(lambda self: -1 * np.ones(self._net.simulation_steps + 1))

    def __repr__(self):
        return f"<{type(self).__name__} {self.link_id}>"


class LinkView(BaseLinkView):
    """Physical link (link.py:30-416); ``is_separator`` selects the Separator behaviour (link.py:418-512)."""

    is_virtual = False

    def __init__(self, network, link_id, start_node, end_node, index, unit_time, is_separator=False, **kw):
        super().__init__(network, link_id, start_node, end_node, index)
        self.is_separator = bool(is_separator)
        self.length = kw["length"]
        self._width = kw["width"]
        self.free_flow_speed = kw["free_flow_speed"]
        self.k_critical = kw["k_critical"]
        self.k_jam = kw["k_jam"]
        self.capacity = self.free_flow_speed * self.k_critical
        self.shockwave_speed = self.capacity / (self.k_jam - self.k_critical)
        self.current_speed = self.free_flow_speed
        self.max_travel_time = self.length / 0.05
        self.bi_factor = kw.get("bi_factor", 1)
        self.fd_type = kw.get("fd_type", "yperman")
        if self.fd_type not in FD_TYPES:
            raise ValueError(f"Unknown model type: {self.fd_type}")
        self.speed_noise_std = kw.get("speed_noise_std", 0)
        self.exponent = 0.8
        self.unit_time = unit_time
        self.gamma = kw.get("gamma", 2e-3)
        self.activity_probability = kw.get("activity_probability", 0.0)
        self.reverse_link = None
        # static derived quantities, evaluated with the reference's own expressions (link.py:83-91,380)
        self.travel_time0 = np.float32(min(self.length / self.free_flow_speed, self.max_travel_time))
        self.free_flow_tau = round(self.travel_time0 / self.unit_time)
        self.avg_travel_time_window = round(100 / self.unit_time)
        self.tau_shockwave = round(self.length / (self.shockwave_speed * self.unit_time))
        # initial widths; the live values are per replica and owned by the network
        w0 = self._width / 2 if self.is_separator else self._width
        self._init_widths = (w0, w0, w0)       # front gate, back gate, separator width

    # --- widths (per replica, write-through) ------------------------------------------------------------
    @property
    def width(self):
        return self._width

    @property
    def front_gate_width(self):
        return self._net._get_width("front", self.index)

    @front_gate_width.setter
    def front_gate_width(self, value: float):
        self._net._set_width("front", self.index, value)
        if self.reverse_link is not None:
            self._net._set_width("back", self.reverse_link.index, value)

    @property
    def back_gate_width(self):
        return self._net._get_width("back", self.index)

    @back_gate_width.setter
    def back_gate_width(self, value: float):
        self._net._set_width("back", self.index, value)
        if self.reverse_link is not None:
            self._net._set_width("front", self.reverse_link.index, value)

    @property
    def separator_width(self):
        if not self.is_separator:
            raise AttributeError("separator_width is only defined for separator links")
        return self._net._get_width("sep", self.index)

    @separator_width.setter
    def separator_width(self, value):
        if not self.is_separator:
            raise AttributeError("separator_width is only defined for separator links")
        net = self._net
        # a numpy float64 width makes density = N / area a binary64 division in the reference (link.py:136,456)
        is_np = 1.0 if isinstance(value, np.floating) else 0.0
        for which in ("sep", "front", "back"):
            net._set_width(which, self.index, value)
        net._set_width("sepnp", self.index, is_np)
        if self.reverse_link is not None:
            for which in ("sep", "front", "back"):
                net._set_width(which, self.reverse_link.index, self._width - value)
            net._set_width("sepnp", self.reverse_link.index, is_np)

    _front_gate_width = property(lambda self: self.front_gate_width)
    _back_gate_width = property(lambda self: self.back_gate_width)
    _separator_width = property(lambda self: self.separator_width)

    @property
    def area(self):
        return self.length * (self.separator_width if self.is_separator else self._width)

    # --- history arrays -----------------------------------------------------------------------------------
    sending_flow = property(lambda self: self._hist("sending_flow"))
    receiving_flow = property(lambda self: self._hist("receiving_flow"))
    back_gate_width_data = property(lambda self: self._hist("back_gate_width_data"))
    travel_time = property(lambda self: self._hist("travel_time"))
    avg_travel_time = property(lambda self: self._hist("avg_travel_time"))
    num_pedestrians = property(lambda self: self._hist("num_pedestrians"))
    density = property(lambda self: self._hist("density"))
    speed = property(lambda self: self._hist("speed"))
    link_flow = property(lambda self: self._hist("link_flow"))

    @property
    def separator_width_data(self):
        if not self.is_separator:
            raise AttributeError("separator_width_data is only defined for separator links")
        # A Separator records its width in both arrays (link.py:451-452); they differ only in the entries no step has
        # written yet: width/2 here (link.py:425) versus the full width in back_gate_width_data (link.py:56).
        rec = np.array(self._hist("back_gate_width_data"), dtype=np.float64)
        t = self._net.current_step
        rec[0] = self._width / 2
        rec[t + 1:] = self._width / 2
        return rec

    def get_density(self, time_step: int):
        """Shared-corridor density (link.py:190-197); a separator reports its own (link.py:427-428)."""
        if self.is_separator:
            return self.density[time_step]
        n = self.num_pedestrians[time_step]
        if self.reverse_link is not None:
            n = n + self.reverse_link.num_pedestrians[time_step]
        return n / np.float32(self.area)

    def get_outflow(self, time_step: int, tau: int):
        """Diffusion outflow of link.py:199-214 from the device histories (the device evaluates it inside the sending flow;
        this is the read-only view for callers that inspect it): the last four inflows from ``tau`` steps ago weighted
        F, F(1-F), F(1-F)^2, F(1-F)^3 with F = 1 / (1 + gamma * avg_travel_time), rounded up, never negative."""
        F = 1 / (1 + self.gamma * self.avg_travel_time[time_step])
        G = 1 - F
        flow = F * self.inflow[time_step - tau]
        for k, weight in ((1, F * G), (2, F * G ** 2), (3, F * G ** 3)):
            flow = flow + weight * self.inflow[time_step - tau - k]
        return max(np.ceil(flow), 0)


class NodeView:
    """Node of the network (node.py:6-26).  ``kind`` is 'one_to_one' or 'regular' (network.py:141-167)."""

    def __init__(self, network, node_id, kind):
        self._net = network
        self.node_id = node_id
        self.kind = kind
        self.incoming_links = []
        self.outgoing_links = []
        self.virtual_incoming_link = None
        self.virtual_outgoing_link = None
        self.source_num = self.dest_num = self.edge_num = None
        self.mask = None
        self.M = 1e6
        self.w = 1e-2
        self.ods_in_turns = {}
        self._demand = None
        self.index = None
        self._tf_set = False

    def _add_virtual_pair(self, vin_index, vout_index):
        self.virtual_incoming_link = BaseLinkView(self._net, f"virtual_in_{self.node_id}", None, self, vin_index)
        self.virtual_outgoing_link = BaseLinkView(self._net, f"virtual_out_{self.node_id}", self, None, vout_index)
        self.incoming_links.append(self.virtual_incoming_link)
        self.outgoing_links.append(self.virtual_outgoing_link)

    def init_node(self):
        self.source_num = len(self.incoming_links)
        self.dest_num = len(self.outgoing_links)
        self.edge_num = self.dest_num * self.source_num - self.source_num
        self.mask = np.ones([self.source_num, self.source_num], dtype=bool)
        np.fill_diagonal(self.mask, False)

    @property
    def demand(self):
        return self._demand

    @demand.setter
    def demand(self, values):
        self._demand = None if values is None else DemandArray(values, on_write=lambda: self._net._mark_demand_dirty(self))
        if values is not None:
            self._net._mark_demand_dirty(self)

    @property
    def turning_fractions(self):
        if not self._tf_set and not self._net._engine_live():
            return None
        return self._net._get_turning_fractions(self)

    @turning_fractions.setter
    def turning_fractions(self, values):
        self._net._set_turning_fractions(self, values)

    def update_matrix_A_eq(self, turning_fractions):
        """Reference entry point for externally imposed fractions (node.py:110-137); only the assignment matters
        for the classic node model."""
        assert len(turning_fractions) == self.edge_num
        self.turning_fractions = turning_fractions

    @property
    def q(self):
        """[outflows of incoming..., inflows of outgoing...] of the latest step (node.py:146-162)."""
        t = self._net.current_step
        if t < 1:
            return None
        return np.array([l.outflow[t] for l in self.incoming_links] + [l.inflow[t] for l in self.outgoing_links])

    def __repr__(self):
        return f"<NodeView {self.node_id} {self.kind} m={self.source_num}>"


class _ReplicaScope:
    """``network.replica(r)``: same ``links``/``nodes`` surface, reads and writes scoped to replica ``r``."""

    def __init__(self, network, r):
        self._net, self._r = network, r

    def __enter__(self):
        self._prev = self._net._scope
        self._net._scope = self._r
        return self._net

    def __exit__(self, *exc):
        self._net._scope = self._prev
        return False


class Network:
    def __init__(self, adjacency_matrix: np.ndarray, params: dict, origin_nodes: list,
                 destination_nodes: list = [], demand_pattern: List[Callable] = None,
                 od_flows: dict = None, pos: dict = None, log_level: int = logging.INFO, verbose: bool = True,
                 n_replicas: int = 1, replica_offset: int = 0, rng_seed: int = 0, rng_mode: str = "philox",
                 device: int = 0, history: str = "full"):
        """``history``: "full" keeps every history array whole like the reference (80 B per link, time index and replica);
        "recent" keeps only what the recurrence looks far back into (include/pedn.h: PEDN_HIST_RECENT) -- same numbers step
        for step, reads of entries that have left their ring raise IndexError."""
        if history not in ("full", "recent"):
            raise ValueError(f"history must be 'full' or 'recent', got {history!r}")
        self.history = history
        self.verbose = verbose
        self.logger = self.setup_logger(log_level) if verbose else None
        self.adjacency_matrix = adjacency_matrix
        self.nodes = {}
        self.links = {}
        self.params = params
        self.simulation_steps = params["simulation_steps"]
        self.unit_time = params["unit_time"]
        self.destination_nodes = destination_nodes
        self.origin_nodes = origin_nodes
        self.path_finder = None
        self.od_manager = None
        self.pos = pos
        self.assign_flows_type = params.get("assign_flows_type", "classic")
        if self.assign_flows_type not in ("classic", "optimal"):
            raise ValueError(f"Invalid type: {self.assign_flows_type}")        # node.py:302
        # 'optimal' (node.py:249-271) is the node LP; the device solves it with its own simplex, which agrees with the reference's
        # scipy/HiGHS in the objective value, not necessarily in the (degenerate) vertex -- see include/pedn.h: node_model
        self.n_replicas = int(n_replicas)
        self.replica_offset = int(replica_offset)
        self.rng_seed, self.rng_mode, self.device = int(rng_seed), rng_mode, int(device)
        self.current_step = 0
        self._scope = None
        self._engine = None
        self._dirty_demand = set()
        self._col_cache = {}
        self._epoch = 0

        self.demand_generator = DemandGenerator(self.simulation_steps, params, self.logger)
        if demand_pattern:
            for func in demand_pattern:
                self.demand_generator.register_pattern(func.__name__, func)

        cc = params.get("controllers", {}) or {}
        self.controller_enabled = cc.get("enabled", False)
        self.controller_nodes = set(map(int, cc.get("nodes", set()) or []))
        self.controller_gaters = self.controller_nodes.copy()
        self.controller_links = cc.get("links", []) or []
        for cl in self.controller_links:
            a, b = cl.split("-")
            self.controller_nodes.add(int(a))
            self.controller_nodes.add(int(b))

        self._vlinks = []            # virtual links in creation order, column = n_links + position
        self._build_nodes_and_links()
        self._init_dynamic_host_state()

        if destination_nodes:
            self.od_manager = ODManager(self.simulation_steps, logger=self.logger)
            self.od_manager.init_od_flows(origin_nodes, destination_nodes, od_flows)
            self.path_finder = PathFinder(self.links, params=self.params, controller_nodes=self.controller_nodes,
                                          controller_links=self.controller_links, logger=self.logger)
            self.path_finder.find_od_paths(od_pairs=self.od_manager.od_flows.keys(), nodes=self.nodes)

    # ------------------------------------------------------------------------------------------ construction
    @staticmethod
    def setup_logger(log_level=logging.INFO, log_dir=None):
        """Console logger (the reference also writes outputs/logs/network.log, network.py:21-54; file logging is
        outside the hot path and omitted)."""
        logger = logging.getLogger("pednstream_amd.network")
        if not logger.handlers:
            h = logging.StreamHandler()
            h.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
            logger.addHandler(h)
            logger.setLevel(log_level)
        return logger

    def _link_params(self, i, j):
        cfg = self.params.get("links", {})
        default = self.params.get("default_link", {})
        for key in (f"{i}_{j}", f"{j}_{i}"):
            if key in cfg:
                return {**default, **cfg[key]}
        return default

    def _make_node(self, nid):
        adj = self.adjacency_matrix
        n_in, n_out = np.sum(adj[:, nid]), np.sum(adj[nid, :])
        is_od = nid in self.origin_nodes or nid in self.destination_nodes
        if n_in == 2 and n_out == 2:
            node = NodeView(self, nid, "regular" if is_od else "one_to_one")
            virtual = is_od
        elif n_in == 1 and n_out == 1:
            node = NodeView(self, nid, "one_to_one")
            virtual = True
        else:
            node = NodeView(self, nid, "regular")
            virtual = is_od
        if virtual:
            self._vlinks.append(node)
            node._pending_virtual = True
            if nid in self.origin_nodes:
                oc = self.params.get("demand", {}).get(f"origin_{nid}", {})
                node._demand = DemandArray(self.demand_generator.generate_custom(nid, oc.get("pattern", "gaussian_peaks")),
                                           on_write=lambda n=node: self._mark_demand_dirty(n))
            else:
                node._demand = DemandArray(np.zeros(self.simulation_steps),
                                           on_write=lambda n=node: self._mark_demand_dirty(n))
        return node

    def _build_nodes_and_links(self):
        adj = self.adjacency_matrix
        n = adj.shape[0]
        pairs = []
        for i in range(n):
            if i not in self.nodes:
                self.nodes[i] = self._make_node(i)
            for j in range(i + 1, n):
                if adj[i, j] == 1:
                    if j not in self.nodes:
                        self.nodes[j] = self._make_node(j)
                    pairs.append((i, j))
        n_links = 2 * len(pairs)
        # virtual pairs sit in front of the physical links of their node (network.py:125-128 runs at node creation)
        for v, node in enumerate(self._vlinks):
            node._add_virtual_pair(n_links + 2 * v, n_links + 2 * v + 1)
        idx = 0
        for (i, j) in pairs:
            lp = self._link_params(i, j)
            if f"{i}-{j}" in self.controller_links or f"{j}-{i}" in self.controller_links:
                kind = "separator"
            else:
                kind = lp.get("controller_type", "gate")
            if kind not in ("separator", "gate"):
                raise ValueError(f"Invalid controller type: {kind}")
            ni, nj = self.nodes[i], self.nodes[j]
            fwd = LinkView(self, f"{i}_{j}", ni, nj, idx, self.unit_time, is_separator=(kind == "separator"), **lp)
            rev = LinkView(self, f"{j}_{i}", nj, ni, idx + 1, self.unit_time, is_separator=(kind == "separator"), **lp)
            idx += 2
            ni.outgoing_links.append(fwd)
            nj.incoming_links.append(fwd)
            ni.incoming_links.append(rev)
            nj.outgoing_links.append(rev)
            self.links[(i, j)] = fwd
            self.links[(j, i)] = rev
            fwd.reverse_link, rev.reverse_link = rev, fwd
        for k, node in enumerate(self.nodes.values()):
            node.index = k
            node.init_node()
        self._link_list = list(self.links.values())
        self.n_links = len(self._link_list)
        self.n_vlinks = 2 * len(self._vlinks)

    def _init_dynamic_host_state(self):
        L, R = self.n_links, self.n_replicas
        init = np.array([l._init_widths for l in self._link_list], dtype=np.float64).reshape(L, 3)
        self._widths = {"front": np.repeat(init[:, 0:1], R, axis=1), "back": np.repeat(init[:, 1:2], R, axis=1),
                        "sep": np.repeat(init[:, 2:3], R, axis=1), "sepnp": np.zeros((L, R))}
        self._tf_host = {}     # node index -> [edge_num, R] array, for values imposed before the engine exists

    # ------------------------------------------------------------------------------------------ engine plumbing
    def _engine_live(self):
        return self._engine is not None

    def engine(self):
        """Create (once) the device engine.  Fails loudly when the HIP library or a GPU is missing."""
        if self._engine is None:
            from . import engine as _engine
            from .flatten import flatten_network

            self._engine = _engine.Engine(flatten_network(self), n_replicas=self.n_replicas,
                                          replica_offset=self.replica_offset, seed=self.rng_seed,
                                          mode=self.rng_mode, device=self.device)
            for which, code in (("front", 0), ("back", 1), ("sep", 2), ("sepnp", 3)):
                self._engine.set_widths(code, self._widths[which])
            for nidx, tf in self._tf_host.items():
                self._engine.set_turning_fractions(nidx, tf)
            # The model description carried the demand arrays as they were when the engine was created, so nothing is
            # dirty now: later host-side writes to ``node.demand`` mark their node (DemandArray.on_write), per-replica
            # uploads through ``set_demand_matrix`` / ``draw_demand`` below clear the mark.
            self._dirty_demand = set()
        return self._engine

    def set_demand_matrix(self, node_id, values):
        """Per-replica demand of one origin, ``values [n_replicas, n]`` in one upload (the reference has one array per run)."""
        node = self.nodes[node_id]
        self._flush().set_demand_matrix(node.index, values)
        self._dirty_demand.discard(node)
        self._invalidate()

    def draw_demand(self, node_id, seed, pattern, base, peak, spike_start, spike_len, spike_height):
        """Per-replica demand of one origin drawn on the device (``pedn_draw_demand``)."""
        node = self.nodes[node_id]
        self._flush().draw_demand(node.index, seed, pattern, base, peak, spike_start, spike_len, spike_height)
        self._dirty_demand.discard(node)
        self._invalidate()

    def _invalidate(self):
        """Cached history columns are keyed by (field, link, replica, step, epoch): anything that can change what a step
        produces -- reset, new demand, new parameters -- starts a new epoch."""
        self._epoch += 1
        self._col_cache.clear()

    def reset(self, lazy=False):
        """Back to t = 0 in place (the reference rebuilds the Network instead, rl/pz_pednet_env.py:163-168): histories cleared
        on the device, ``current_step`` 0, no cached column survives.  ``lazy``: see ``Engine.reset`` (the batched RL env resets
        this way: 2.9 -> 0.4 ms for 19.5 GB of histories)."""
        self._flush().reset(lazy=lazy)
        self.current_step = 0
        self._invalidate()

    def _flush(self):
        eng = self.engine()
        if self._dirty_demand:
            for node in self._dirty_demand:
                eng.set_demand(node.index, np.asarray(node._demand, dtype=np.float64), None)
            self._dirty_demand = set()
            self._invalidate()
        return eng

    def _mark_demand_dirty(self, node):
        self._dirty_demand.add(node)

    def _replica_index(self):
        return 0 if self._scope is None else self._scope

    def _read_column(self, fid, link_index):
        eng = self._flush()
        key = (fid, link_index, self._replica_index(), self.current_step, self._epoch)
        col = self._col_cache.get(key)
        if col is None:
            if len(self._col_cache) > 4096:
                self._col_cache.clear()
            rows = eng.history_rows(fid)
            if rows < self.simulation_steps + 1:      # recent-history mode: the entries still in the ring, NaN elsewhere
                # sending_flow / receiving_flow of step t are entries t - 1: their newest entry is one behind (in full-record
                # mode entry t still holds its initial -1 at that point; a ring slot would hold the value of t - rows)
                hi = max(self.current_step - 1 if fid in (LINK_FIELDS["sending_flow"][0], LINK_FIELDS["receiving_flow"][0]) else self.current_step, 0)
                lo = max(0, hi - rows + 1)
                col = np.full(self.simulation_steps + 1, np.nan, dtype=LINK_DTYPES[fid])
                col[lo:hi + 1] = eng.read_block(fid, lo, hi + 1, link_index, link_index + 1, self._replica_index(),
                                                self._replica_index() + 1).reshape(-1)
            else:
                col = eng.read_column(fid, link_index, self._replica_index())
            col.setflags(write=False)
            self._col_cache[key] = col
        return col

    def _read_element(self, fid, link_index, t):
        key = (fid, link_index, self._replica_index(), self.current_step, self._epoch)
        col = self._col_cache.get(key)
        if col is not None:
            return col[t]
        try:
            return self._flush().read_element(fid, link_index, self._replica_index(), t)
        except RuntimeError as err:
            if "ring" in str(err):
                raise IndexError(str(err)) from None
            raise

    def _refresh_widths(self):
        """Pull the widths back after the device changed them (batched RL actions)."""
        if getattr(self, "_widths_stale", False) and self._engine is not None:
            for which, code in (("front", 0), ("back", 1), ("sep", 2), ("sepnp", 3)):
                self._widths[which] = self._engine.get_widths(code)
        self._widths_stale = False

    def _get_width(self, which, link_index):
        self._refresh_widths()
        return float(self._widths[which][link_index, self._replica_index()])

    def _set_width(self, which, link_index, value):
        code = {"front": 0, "back": 1, "sep": 2, "sepnp": 3}[which]
        self._refresh_widths()
        if self._scope is None:
            self._widths[which][link_index, :] = value
        else:
            self._widths[which][link_index, self._scope] = value
        if self._engine is not None:
            self._engine.set_width(code, link_index, float(value), self._scope)

    def _get_turning_fractions(self, node):
        if self._engine is None:
            tf = self._tf_host.get(node.index)
            return None if tf is None else tf[:, self._replica_index()].copy()
        return self._engine.get_turning_fractions(node.index, self._replica_index())

    def _set_turning_fractions(self, node, values):
        values = np.asarray(values, dtype=np.float64).reshape(-1)
        if len(values) != node.edge_num:
            raise ValueError(f"node {node.node_id}: expected {node.edge_num} turning fractions, got {len(values)}")
        node._tf_set = True
        if self._engine is None:
            tf = self._tf_host.setdefault(node.index, np.tile(np.full(node.edge_num, np.nan)[:, None], (1, self.n_replicas)))
            if self._scope is None:
                tf[:, :] = values[:, None]
            else:
                tf[:, self._scope] = values
        else:
            self._engine.set_turning_fractions(node.index, values, self._scope)

    # ------------------------------------------------------------------------------------------ public API
    def replica(self, r: int):
        """Context manager scoping link/node reads and writes to replica ``r``."""
        if not 0 <= r < self.n_replicas:
            raise IndexError("replica out of range")
        return _ReplicaScope(self, int(r))

    def update_turning_fractions_per_node(self, node_ids: List[int], new_turning_fractions: np.ndarray):
        for i, n in enumerate(node_ids):
            self.nodes[n].update_matrix_A_eq(new_turning_fractions[i])

    def network_loading(self, time_step: int):
        """One LTM step for every replica (network.py:266-287).  ``time_step`` runs from 1 to T-1."""
        if not 1 <= time_step <= self.simulation_steps:
            raise IndexError(f"time_step {time_step} outside 1..{self.simulation_steps}")
        eng = self._flush()
        eng.step(time_step)
        self.current_step = time_step

    def run(self, t0: int, t1: int, check: bool = True):
        """Steps ``t0 .. t1-1`` enqueued back to back without host synchronisation in between."""
        eng = self._flush()
        eng.run(t0, t1)
        self.current_step = t1 - 1
        if check:
            eng.check_errors()

    def update_link_states(self, time_step: int):
        """Part of ``network_loading`` on the device (network.py:257-264); kept for interface completeness."""
        raise NotImplementedError("link states are updated inside network_loading on the device")

    def read_field(self, name: str, t0: int = 0, t1: int = None) -> np.ndarray:
        """Bulk device->host copy of one history field: ``[t1-t0, n_links(+virtual), n_replicas]``."""
        fid, _ = LINK_FIELDS[name]
        t1 = self.simulation_steps + 1 if t1 is None else t1
        return self._flush().read_block(fid, t0, t1)

    def synchronize(self):
        self._flush().synchronize()

    def close(self):
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    def visualize(self, *a, **k):
        """The reference draws the topology with networkx / matplotlib here (network.py:289-351).  Plotting is outside the hot path and
        not provided; the reference's examples call this right after construction (examples/nine_node.py:82, long_corridor.py:120), so
        it warns and returns instead of raising -- the rest of such a script runs."""
        import warnings

        warnings.warn("Network.visualize: plotting is not provided by pednstream_amd (outside the network_loading hot path)", stacklevel=2)
        return None
