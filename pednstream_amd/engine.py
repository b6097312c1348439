"""ctypes binding of the C-ABI engine library (include/pedn.h -> pednstream_amd/csrc/libpedn_hip.so).

There is deliberately no CPU fallback: if the HIP library is missing or no GPU is visible, creating an
``Engine`` raises.  The scalar CPU restatement under oracle/ is test infrastructure and is never imported here.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PEDN_HIP_LIB") or os.path.join(_HERE, "csrc", "libpedn_hip.so")   # env override: A/B builds

PEDN_ALL = -1
ABI_VERSION = 4
ERROR_BITS = {1: "negative sending flow (ValueError, link.py:345-346,365-366)",
              2: "negative flows at a node (Warning, node.py:192-194,218-219,237-238)",
              4: "history index out of range (IndexError)",
              8: "binomial with n < 0 (ValueError, link.py:382)",
              16: "zero-step look-back: result depends on node iteration order in the reference",
              32: "node LP (assign_flows_type 'optimal') did not terminate"}

_I32P, _F64P, _F32P = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_float)


class ModelDesc(C.Structure):
    """Mirror of ``pedn_model_desc`` (include/pedn.h); field order and types must match the header."""

    _fields_ = [
        ("abi_version", C.c_int32),
        ("n_nodes", C.c_int32), ("n_links", C.c_int32), ("n_vlinks", C.c_int32), ("n_turns", C.c_int32),
        ("n_demand", C.c_int32), ("n_od", C.c_int32),
        ("T", C.c_int32), ("window", C.c_int32), ("dt", C.c_double),
        ("node_kind", _I32P), ("node_slot_ptr", _I32P), ("node_turn_ptr", _I32P), ("node_demand_row", _I32P),
        ("node_dyn", _I32P), ("slot_in_link", _I32P), ("slot_out_link", _I32P),
        ("link_rev", _I32P), ("link_sep", _I32P), ("link_fd", _I32P), ("link_tau_sw", _I32P), ("link_fft", _I32P),
        ("link_tt0", _F32P),
        ("link_length", _F64P), ("link_width", _F64P), ("link_vf", _F64P), ("link_kc", _F64P), ("link_kj", _F64P),
        ("link_gamma", _F64P), ("link_act", _F64P), ("link_bi", _F64P), ("link_noise", _F64P),
        ("front_gate0", _F64P), ("back_gate0", _F64P), ("sep_width0", _F64P),
        ("tf_init", _F64P), ("demand", _F64P), ("od_w", _F64P),
        ("pf_temp", C.c_double), ("pf_alpha", C.c_double), ("pf_beta", C.c_double), ("pf_omega", C.c_double),
        ("pf_eps", C.c_double),
        ("n_up", C.c_int32), ("n_upod", C.c_int32), ("n_grp", C.c_int32), ("n_ent", C.c_int32), ("n_pair", C.c_int32),
        ("node_up_ptr", _I32P), ("up_slot", _I32P), ("up_od_ptr", _I32P), ("upod_od", _I32P), ("node_grp_ptr", _I32P),
        ("grp_ent_ptr", _I32P), ("grp_allphys", _I32P), ("grp_node", _I32P), ("ent_link", _I32P), ("ent_dist", _F64P),
        ("turn_pair_ptr", _I32P), ("pair_ent", _I32P), ("pair_upod", _I32P),
        ("history_mode", C.c_int32), ("node_model", C.c_int32),
        ("node_cost", _F32P),
    ]


class RlDesc(C.Structure):
    """Mirror of ``pedn_rl_desc`` (include/pedn.h)."""

    _fields_ = [("n_agents", C.c_int32), ("agent_type", _I32P), ("agent_link_ptr", _I32P), ("agent_links", _I32P),
                ("obs_mode", C.c_int32), ("normalize", C.c_int32), ("reward_mode", C.c_int32),
                ("max_delta_sep", C.c_double), ("max_delta_gate", C.c_double), ("min_sep", C.c_double)]


_PTR_TYPES = {_I32P: np.int32, _F64P: np.float64, _F32P: np.float32}


def build_model_desc(model: dict):
    """dict from ``flatten_network`` -> (ModelDesc, keepalive list).  The arrays are borrowed by the struct."""
    desc = ModelDesc()
    keep = []
    for name, ctype in ModelDesc._fields_:
        if name == "abi_version":
            desc.abi_version = ABI_VERSION
        elif name == "node_cost" and model.get("node_cost") is None:
            desc.node_cost = None                                 # optional: no measured packing cost for this scenario
        elif ctype in _PTR_TYPES:
            arr = np.ascontiguousarray(model[name], dtype=_PTR_TYPES[ctype])
            if arr.size == 0:
                arr = np.zeros(1, dtype=_PTR_TYPES[ctype])      # never hand out NULL for an empty table
            keep.append(arr)
            setattr(desc, name, arr.ctypes.data_as(ctype))
        else:
            setattr(desc, name, model.get(name, 0) if name in ("history_mode", "node_model") else model[name])
    return desc, keep


def _bind_hip_runtime():
    """One HIP runtime per process, whatever the import order.  PyTorch-ROCm ships its own ``libamdhip64.so`` (soname
    ``libamdhip64.so.7``, found through torch's RPATH by FILE name); the engine library asks for the soname.  Engine first
    used to mean: the system runtime is loaded for the engine, torch then loads its private copy next to it, and the second
    runtime sees no GPU.  So, when torch is installed, its copy is opened here first with RTLD_GLOBAL -- the engine's
    ``DT_NEEDED libamdhip64.so.7`` then resolves to that already loaded object, and a later ``import torch`` finds its file
    loaded too.  torch is NOT imported.  ``PEDN_HIP_RUNTIME=system`` keeps the system runtime (torch must then not be used in
    this process with the GPU)."""
    if os.environ.get("PEDN_HIP_RUNTIME", "auto") == "system":
        return None
    import importlib.util

    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


_hip_runtime = None


def _load():
    global _hip_runtime
    if not os.path.exists(LIB_PATH):
        # build on demand when the toolchain is there (same recipe as __graft_entry__.build()); never a CPU fallback
        import shutil
        import subprocess

        if shutil.which("make") and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
            subprocess.run(["make", "-s", "-C", os.path.join(_HERE, "csrc"), "libpedn_hip.so"], check=False)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"HIP engine library not built: {LIB_PATH} is missing. Run `python -c 'import "
                           f"__graft_entry__ as g; g.build()'` or `make -C pednstream_amd/csrc` (needs hipcc). "
                           f"There is no CPU fallback for the product path.")
    _hip_runtime = _bind_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    P = C.c_void_p
    sig = {
        "pedn_abi_version": (C.c_int, []),
        "pedn_last_error": (C.c_char_p, [P]),
        "pedn_create": (C.c_int, [C.POINTER(ModelDesc), C.c_int32, C.c_int32, C.c_uint64, C.c_int32, C.c_int32,
                                  C.POINTER(P)]),
        "pedn_destroy": (C.c_int, [P]),
        "pedn_set_demand": (C.c_int, [P, C.c_int32, C.c_int32, _F64P, C.c_int32]),
        "pedn_set_demand_matrix": (C.c_int, [P, C.c_int32, _F64P, C.c_int32]),
        "pedn_set_demand_rows": (C.c_int, [P, C.c_int32, _I32P, C.c_int32, _F64P, C.c_int32]),
        "pedn_get_demand": (C.c_int, [P, C.c_int32, C.c_int32, _F64P, C.c_int32]),
        "pedn_draw_demand": (C.c_int, [P, C.c_int32, C.c_uint64, _I32P, _F64P, _F64P, _I32P, _I32P, _F64P]),
        "pedn_set_od_weights": (C.c_int, [P, C.c_int32, _F64P, C.c_int32]),
        "pedn_set_turning_fractions": (C.c_int, [P, C.c_int32, C.c_int32, _F64P, C.c_int32]),
        "pedn_get_turning_fractions": (C.c_int, [P, C.c_int32, C.c_int32, _F64P, C.c_int32]),
        "pedn_set_width": (C.c_int, [P, C.c_int32, C.c_int32, C.c_int32, C.c_double]),
        "pedn_set_widths": (C.c_int, [P, C.c_int32, _F64P]),
        "pedn_get_widths": (C.c_int, [P, C.c_int32, _F64P]),
        "pedn_reset_widths": (C.c_int, [P, _F64P, _F64P, _F64P]),
        "pedn_step": (C.c_int, [P, C.c_int32]),
        "pedn_run": (C.c_int, [P, C.c_int32, C.c_int32]),
        "pedn_synchronize": (C.c_int, [P]),
        "pedn_error_flags": (C.c_int, [P, C.POINTER(C.c_uint32)]),
        "pedn_read": (C.c_int, [P, C.c_int32] + [C.c_int32] * 6 + [C.c_void_p]),
        "pedn_device_ptr": (C.c_void_p, [P, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
        "pedn_history_rows": (C.c_int, [P, C.c_int32]),
        "pedn_stream": (C.c_void_p, [P]),
        "pedn_timer_begin": (C.c_int, [P]),
        "pedn_timer_end": (C.c_int, [P, C.POINTER(C.c_float)]),
        "pedn_reset": (C.c_int, [P]),
        "pedn_reset_lazy": (C.c_int, [P]),
        "pedn_profile_step": (C.c_int, [P, C.c_int32, C.POINTER(C.c_float)]),
        "pedn_profile_run": (C.c_int, [P, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
        "pedn_profile_timeline": (C.c_int, [P, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "pedn_set_streams": (C.c_int, [P, C.c_int32]),
        "pedn_plan_info": (C.c_int, [P, _I32P, C.c_int32]),
        "pedn_set_link_params": (C.c_int, [P, _F64P, _F64P, _F64P, _I32P, _I32P, _F32P]),
        "pedn_set_od_weights_per_replica": (C.c_int, [P, _F64P]),
        "pedn_get_od_weights_per_replica": (C.c_int, [P, _F64P]),
        "pedn_get_link_params": (C.c_int, [P, _F64P, _F64P, _F64P, _I32P, _I32P, _F32P]),
        "pedn_randomize_scenarios": (C.c_int, [P, C.c_uint64, C.c_double, C.c_int32, _I32P, C.c_int32]),
        "pedn_rl_configure": (C.c_int, [P, C.POINTER(RlDesc), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "pedn_rl_apply_actions": (C.c_int, [P, C.c_void_p, C.c_int32]),
        "pedn_rl_observe": (C.c_int, [P, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
        "pedn_rl_fetch": (C.c_int, [P, C.c_void_p, C.c_void_p]),
        "pedn_rl_step_many": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
        "pedn_rl_step": (C.c_int, [P, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
        "pedn_rl_device_ptr": (C.c_void_p, [P, C.c_int32]),
        "pedn_rl_clock_begin": (C.c_int, [P, C.c_int32]),
        "pedn_rl_step_clocked": (C.c_int, [P, C.c_void_p, C.c_int32, C.c_void_p]),
        "pedn_rl_clock_end": (C.c_int, [P, C.POINTER(C.c_int32)]),
        "pedn_rl_clocked": (C.c_int, [P]),
        "pedn_rl_clock_signature": (C.c_uint64, [P]),
        "pedn_flush": (C.c_int, [P]),
        "pedn_device_math": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _F64P, _F64P, C.c_uint64, _F64P]),
    }
    # the version first: a stale or alternate library (PEDN_HIP_LIB) must fail with this message, not with an AttributeError on a symbol
    lib.pedn_abi_version.restype, lib.pedn_abi_version.argtypes = C.c_int, []
    if lib.pedn_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{LIB_PATH}: ABI version {lib.pedn_abi_version()}, this package needs {ABI_VERSION} -- rebuild with "
                           f"`make -C pednstream_amd/csrc`")
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


EXPORTS = ["pedn_abi_version", "pedn_last_error", "pedn_create", "pedn_destroy", "pedn_set_demand", "pedn_set_demand_matrix", "pedn_set_demand_rows", "pedn_get_demand", "pedn_draw_demand",
           "pedn_set_od_weights", "pedn_set_turning_fractions", "pedn_get_turning_fractions", "pedn_set_width",
           "pedn_set_widths", "pedn_step", "pedn_run", "pedn_synchronize", "pedn_error_flags", "pedn_read",
           "pedn_device_ptr", "pedn_history_rows", "pedn_stream", "pedn_timer_begin", "pedn_timer_end", "pedn_reset", "pedn_reset_lazy", "pedn_device_math", "pedn_profile_step", "pedn_profile_run", "pedn_profile_timeline", "pedn_set_streams", "pedn_plan_info", "pedn_rl_configure",
           "pedn_rl_apply_actions", "pedn_rl_observe", "pedn_rl_fetch", "pedn_rl_step_many", "pedn_rl_step", "pedn_rl_device_ptr", "pedn_get_widths", "pedn_set_link_params",
           "pedn_set_od_weights_per_replica", "pedn_get_od_weights_per_replica", "pedn_get_link_params", "pedn_randomize_scenarios", "pedn_reset_widths",
           "pedn_flush", "pedn_rl_clock_begin", "pedn_rl_step_clocked", "pedn_rl_clock_end", "pedn_rl_clocked", "pedn_rl_clock_signature"]


def _p(a, dtype=np.float64):
    """ctypes pointer to a contiguous numpy array of the given element type."""
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as({np.float64: _F64P, np.int32: _I32P, np.float32: _F32P}[dtype])


class ModelError(RuntimeError):
    """A replica hit one of the reference's raise sites; ``flags`` holds the per-replica bit masks."""

    def __init__(self, msg, flags):
        super().__init__(msg)
        self.flags = flags


class Engine:
    """One handle = one scenario x ``n_replicas`` on one GPU."""

    def __init__(self, model: dict, n_replicas=1, replica_offset=0, seed=0, mode="philox", device=0):
        self._lib = lib()
        self.model = model
        self.n_replicas = int(n_replicas)
        self.T = int(model["T"])
        self.n_links = int(model["n_links"])
        self.n_all = self.n_links + int(model["n_vlinks"])
        desc, self._keep = build_model_desc(model)
        h = C.c_void_p()
        rc = self._lib.pedn_create(C.byref(desc), self.n_replicas, int(replica_offset), int(seed),
                                   {"philox": 0, "meanfield": 1}[mode], int(device), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"pedn_create failed ({rc}): {self._lib.pedn_last_error(None).decode()}")
        self._h = h

    # -- helpers
    def _ck(self, rc):
        if rc < 0:
            raise RuntimeError(f"pedn call failed ({rc}): {self._lib.pedn_last_error(self._h).decode()}")
        return rc

    @staticmethod
    def _rep(replica):
        return PEDN_ALL if replica is None else int(replica)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pedn_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- setters
    def set_demand(self, node_index, values, replica=None):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._ck(self._lib.pedn_set_demand(self._h, int(node_index), self._rep(replica), v.ctypes.data_as(_F64P), len(v)))

    def get_demand(self, node_index, replica=0, n=None):
        out = np.empty(self.T + 1 if n is None else int(n), dtype=np.float64)
        self._ck(self._lib.pedn_get_demand(self._h, int(node_index), int(replica), out.ctypes.data_as(_F64P), len(out)))
        return out

    def set_demand_matrix(self, node_index, values):
        """values [n_replicas, n]: the origin's demand of every replica in one upload."""
        v = np.ascontiguousarray(values, dtype=np.float64)
        if v.ndim != 2 or v.shape[0] != self.n_replicas:
            raise ValueError(f"expected [n_replicas={self.n_replicas}, n], got {v.shape}")
        self._ck(self._lib.pedn_set_demand_matrix(self._h, int(node_index), v.ctypes.data_as(_F64P), v.shape[1]))

    def set_demand_rows(self, node_index, replicas, values):
        """values [len(replicas), n]: the origin's demand of the listed replicas in one upload."""
        rep = np.ascontiguousarray(replicas, dtype=np.int32)
        v = np.ascontiguousarray(values, dtype=np.float64)
        if v.ndim != 2 or v.shape[0] != len(rep):
            raise ValueError(f"expected [{len(rep)}, n], got {v.shape}")
        self._ck(self._lib.pedn_set_demand_rows(self._h, int(node_index), rep.ctypes.data_as(_I32P), len(rep), v.ctypes.data_as(_F64P), v.shape[1]))

    def draw_demand(self, node_index, seed, pattern, base, peak, spike_start, spike_len, spike_height):
        """Origin demand of every replica drawn on the device (include/pedn.h: pedn_draw_demand); arrays [n_replicas]."""
        R = self.n_replicas
        i32 = [np.ascontiguousarray(a, dtype=np.int32) for a in (pattern, spike_start, spike_len)]
        f64 = [np.ascontiguousarray(a, dtype=np.float64) for a in (base, peak, spike_height)]
        if any(a.shape != (R,) for a in i32 + f64):
            raise ValueError(f"every argument must have shape [{R}]")
        self._ck(self._lib.pedn_draw_demand(self._h, int(node_index), int(seed) & (2 ** 64 - 1), i32[0].ctypes.data_as(_I32P),
                                            f64[0].ctypes.data_as(_F64P), f64[1].ctypes.data_as(_F64P), i32[1].ctypes.data_as(_I32P),
                                            i32[2].ctypes.data_as(_I32P), f64[2].ctypes.data_as(_F64P)))

    def set_od_weights(self, od_index, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self._ck(self._lib.pedn_set_od_weights(self._h, int(od_index), v.ctypes.data_as(_F64P), len(v)))

    def set_turning_fractions(self, node_index, tf, replica=None):
        tf = np.asarray(tf, dtype=np.float64)
        if tf.ndim == 2:       # [edge_num, R] host mirror: NaN rows mean "not imposed"
            if not np.isnan(tf).any() and (tf == tf[:, :1]).all():
                return self.set_turning_fractions(node_index, tf[:, 0], None)     # same for every replica: keeps the scalar fast path
            for r in range(tf.shape[1]):
                if not np.isnan(tf[:, r]).any():
                    self.set_turning_fractions(node_index, tf[:, r], r)
            return
        v = np.ascontiguousarray(tf)
        self._ck(self._lib.pedn_set_turning_fractions(self._h, int(node_index), self._rep(replica),
                                                      v.ctypes.data_as(_F64P), len(v)))

    def get_turning_fractions(self, node_index, replica=0):
        a, b = self.model["node_turn_ptr"][node_index], self.model["node_turn_ptr"][node_index + 1]
        out = np.empty(int(b - a), dtype=np.float64)
        self._ck(self._lib.pedn_get_turning_fractions(self._h, int(node_index), int(replica), out.ctypes.data_as(_F64P), len(out)))
        return out

    def set_width(self, which, link, value, replica=None):
        self._ck(self._lib.pedn_set_width(self._h, int(which), int(link), self._rep(replica), float(value)))

    def set_widths(self, which, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        assert v.shape == (self.n_links, self.n_replicas)
        self._ck(self._lib.pedn_set_widths(self._h, int(which), v.ctypes.data_as(_F64P)))

    def reset_widths(self, front, back, sep):
        """Every replica back to the widths front / back / sep [n_links] (an episode reset), float64-separator marks cleared."""
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (front, back, sep)]
        assert all(x.shape == (self.n_links,) for x in a)
        self._ck(self._lib.pedn_reset_widths(self._h, _p(a[0]), _p(a[1]), _p(a[2])))

    def set_link_params(self, kc, kj, vf, fft, tau_sw, tt0):
        """Per-replica k_critical / k_jam / free_flow_speed [n_links, n_replicas] + host-derived look-backs; None resets."""
        if kc is None:
            self._ck(self._lib.pedn_set_link_params(self._h, None, None, None, None, None, None))
            return
        arrs = [np.ascontiguousarray(a, dtype=dt) for a, dt in ((kc, np.float64), (kj, np.float64), (vf, np.float64),
                                                                  (fft, np.int32), (tau_sw, np.int32), (tt0, np.float32))]
        for a in arrs:
            assert a.shape == (self.n_links, self.n_replicas), a.shape
        self._ck(self._lib.pedn_set_link_params(self._h, arrs[0].ctypes.data_as(_F64P), arrs[1].ctypes.data_as(_F64P),
                                                arrs[2].ctypes.data_as(_F64P), arrs[3].ctypes.data_as(_I32P),
                                                arrs[4].ctypes.data_as(_I32P), arrs[5].ctypes.data_as(_F32P)))

    def set_od_weights_per_replica(self, w):
        if w is None:
            self._ck(self._lib.pedn_set_od_weights_per_replica(self._h, None))
            return
        w = np.ascontiguousarray(w, dtype=np.float64)
        assert w.shape == (int(self.model["n_od"]), self.n_replicas), w.shape
        self._ck(self._lib.pedn_set_od_weights_per_replica(self._h, w.ctypes.data_as(_F64P)))

    def get_link_params(self):
        """The per-replica link parameters as they are on the device: dict of [n_links, n_replicas] arrays."""
        L, R = self.n_links, self.n_replicas
        out = {"kc": np.empty((L, R)), "kj": np.empty((L, R)), "vf": np.empty((L, R)), "fft": np.empty((L, R), dtype=np.int32),
               "tau_sw": np.empty((L, R), dtype=np.int32), "tt0": np.empty((L, R), dtype=np.float32)}
        self._ck(self._lib.pedn_get_link_params(self._h, _p(out["kc"]), _p(out["kj"]), _p(out["vf"]), _p(out["fft"], np.int32),
                                                _p(out["tau_sw"], np.int32), _p(out["tt0"], np.float32)))
        return out

    def get_od_weights_per_replica(self):
        w = np.empty((int(self.model["n_od"]), self.n_replicas))
        self._ck(self._lib.pedn_get_od_weights_per_replica(self._h, _p(w)))
        return w

    def randomize_scenarios(self, seed, link_fraction=0.2, links=True, od_weights=True, origin_nodes=()):
        """pedn_randomize_scenarios: every replica's scenario drawn on the device (origin_nodes: model node indices whose demand is drawn)."""
        nodes = np.ascontiguousarray(origin_nodes, dtype=np.int32)
        what = (1 if links else 0) | (2 if od_weights else 0) | (4 if len(nodes) else 0)
        self._ck(self._lib.pedn_randomize_scenarios(self._h, int(seed) & (2 ** 64 - 1), float(link_fraction), what, _p(nodes, np.int32), len(nodes)))

    def get_widths(self, which):
        out = np.empty((self.n_links, self.n_replicas), dtype=np.float64)
        self._ck(self._lib.pedn_get_widths(self._h, int(which), out.ctypes.data_as(_F64P)))
        return out

    # -- stepping
    def step(self, t):
        self._ck(self._lib.pedn_step(self._h, int(t)))

    def run(self, t0, t1):
        self._ck(self._lib.pedn_run(self._h, int(t0), int(t1)))

    def synchronize(self):
        self._ck(self._lib.pedn_synchronize(self._h))

    def reset(self, lazy=False):
        """Back to t = 0.  lazy: only what a new episode reads before it writes is restored (pedn_reset_lazy); rows of steps that have
        not run yet read as their initial values through read_block / read_column, but a zero-copy consumer (device_ptr) makes the
        engine clear them after all."""
        self._ck(self._lib.pedn_reset_lazy(self._h) if lazy else self._lib.pedn_reset(self._h))

    def error_flags(self):
        flags = np.zeros(self.n_replicas, dtype=np.uint32)
        rc = self._ck(self._lib.pedn_error_flags(self._h, flags.ctypes.data_as(C.POINTER(C.c_uint32))))
        return rc, flags

    def check_errors(self):
        """Raise the reference's exception type for the first sticky error bit found (SURVEY 8b error conventions)."""
        rc, flags = self.error_flags()
        if rc == 0:
            return
        bad = int(np.flatnonzero(flags)[0])
        bits = int(flags[bad])
        msg = "; ".join(text for bit, text in ERROR_BITS.items() if bits & bit)
        msg = f"replica {bad}: {msg}"
        if bits & (1 | 8):
            err = ValueError(msg)
        elif bits & 4:
            err = IndexError(msg)
        else:
            err = ModelError(msg, flags)
        err.flags = flags
        raise err

    def profile_step(self, t):
        """One step with per-kernel HIP-event timing: (turn_prob_ms, node_ms, link_ms)."""
        ms = (C.c_float * 3)()
        self._ck(self._lib.pedn_profile_step(self._h, int(t), ms))
        return tuple(float(x) for x in ms)

    def profile_run(self, t0, t1):
        """Steps t0 <= t < t1 under run()'s launch plan with every launch timed: ((turn_ms, node_ms, second_ms) mean per launch,
        chains) -- chains = 2 when the two halves of the replicas ran as two chains of launches on two streams."""
        ms = (C.c_float * 3)()
        chains = C.c_int32(0)
        self._ck(self._lib.pedn_profile_run(self._h, int(t0), int(t1), ms, C.byref(chains)))
        return tuple(float(x) for x in ms), int(chains.value)

    def profile_timeline(self, t0, t1):
        """The launches of steps t0 <= t < t1 one by one: (rows [n, 5] = step, chain, kind (0 stand-alone turning fractions, 1 node
        kernel, 2 the launch behind it), start ms, end ms after the first launch's start; chains)."""
        cap = 6 * (int(t1) - int(t0))
        buf = np.zeros(5 * cap, dtype=np.float32)
        n, chains = C.c_int32(0), C.c_int32(0)
        self._ck(self._lib.pedn_profile_timeline(self._h, int(t0), int(t1), buf.ctypes.data_as(C.POINTER(C.c_float)), cap, C.byref(n), C.byref(chains)))
        return buf[:5 * n.value].reshape(-1, 5).astype(np.float64), int(chains.value)

    def plan_info(self):
        """The launch plan of run(): chains, owner-wave link update, and the result of the stream-overlap probe (include/pedn.h)."""
        info = np.zeros(5, dtype=np.int32)
        self._ck(self._lib.pedn_plan_info(self._h, info.ctypes.data_as(_I32P), 5))
        return {"chains": int(info[0]), "link_update_by_next_node_kernel": bool(info[1]), "stream_probe_attempts": int(info[2]),
                "stream_probe_us": int(info[3]), "packed_by": ("degree", "static_load_estimate", "measured_node_cost")[int(info[4])]}

    def set_streams(self, n):
        """Launch plan of run() for long ranges: 1 chain of launches, or 2 (the halves of the replica batch on two streams; falls
        back to one when the runtime's hardware queues do not run them side by side -- see plan_info())."""
        self._ck(self._lib.pedn_set_streams(self._h, int(n)))

    # -- batched RL glue
    def rl_configure(self, agent_type, agent_link_ptr, agent_links, obs_mode, normalize, reward_mode, max_delta_sep,
                     max_delta_gate, min_sep):
        keep = [np.ascontiguousarray(a, dtype=np.int32) for a in (agent_type, agent_link_ptr, agent_links)]
        d = RlDesc(len(keep[0]), keep[0].ctypes.data_as(_I32P), keep[1].ctypes.data_as(_I32P), keep[2].ctypes.data_as(_I32P),
                   int(obs_mode), int(bool(normalize)), int(reward_mode), float(max_delta_sep), float(max_delta_gate), float(min_sep))
        na, no = C.c_int32(), C.c_int32()
        rc = self._lib.pedn_rl_configure(self._h, C.byref(d), C.byref(na), C.byref(no))
        if rc < 0:
            msg = self._lib.pedn_last_error(self._h).decode()
            raise (IndexError if "IndexError" in msg else ValueError)(msg)
        self.rl_n_agents, self.rl_n_actions, self.rl_n_obs = len(keep[0]), na.value, no.value
        return na.value, no.value

    def rl_apply_actions(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float64)
        assert a.shape == (self.n_replicas, self.rl_n_actions), a.shape
        self._ck(self._lib.pedn_rl_apply_actions(self._h, a.ctypes.data_as(C.c_void_p), 0))

    def rl_observe(self, t, accumulate=False, fetch=True):
        if not fetch:
            self._ck(self._lib.pedn_rl_observe(self._h, int(t), int(accumulate), None, None))
            return None, None
        obs = np.empty((self.n_replicas, self.rl_n_obs), dtype=np.float32)
        rew = np.empty((self.n_replicas, self.rl_n_agents), dtype=np.float32)
        self._ck(self._lib.pedn_rl_observe(self._h, int(t), int(accumulate), obs.ctypes.data_as(C.c_void_p), rew.ctypes.data_as(C.c_void_p)))
        return obs, rew

    @staticmethod
    def rl_step_many(engines, actions, t, action_gap=1):
        """pedn_rl_step_many: one env step of SEVERAL engines (same step index, same agent layout) and their observations / rewards in
        one library call; rows are the engines' replicas one after the other."""
        e0 = engines[0]
        n_rows = sum(e.n_replicas for e in engines)
        handles = (C.c_void_p * len(engines))(*[e._h for e in engines])
        a = None if actions is None else np.ascontiguousarray(actions, dtype=np.float64)
        if a is not None:
            assert a.shape == (n_rows, e0.rl_n_actions), a.shape
        obs = np.empty((n_rows, e0.rl_n_obs), dtype=np.float32)
        rew = np.empty((n_rows, e0.rl_n_agents), dtype=np.float32)
        rc = e0._lib.pedn_rl_step_many(handles, len(engines), None if a is None else a.ctypes.data_as(C.c_void_p), int(t), int(action_gap),
                                       obs.ctypes.data_as(C.c_void_p), rew.ctypes.data_as(C.c_void_p))
        if rc != 0:
            raise RuntimeError(f"pedn_rl_step_many failed ({rc}): {e0._lib.pedn_last_error(None).decode()}")
        return obs, rew

    def rl_fetch(self):
        """Host copies of the observation / reward buffers as the last rl_step / rl_observe left them (pedn_rl_fetch)."""
        obs = np.empty((self.n_replicas, self.rl_n_obs), dtype=np.float32)
        rew = np.empty((self.n_replicas, self.rl_n_agents), dtype=np.float32)
        self._ck(self._lib.pedn_rl_fetch(self._h, obs.ctypes.data_as(C.c_void_p), rew.ctypes.data_as(C.c_void_p)))
        return obs, rew

    def rl_step(self, actions, t, action_gap=1, fetch=True, ordered=False):
        """apply -> action_gap x (step, observe); returns host copies of the last observations and the summed rewards.
        ordered (only without actions and without a fetch): every launch on ``stream_ptr()``, see rl_step_device."""
        if ordered:
            assert actions is None and not fetch
            self._ck(self._lib.pedn_rl_step(self._h, None, 2, int(t), int(action_gap), None, None))
            return None, None
        a = None if actions is None else np.ascontiguousarray(actions, dtype=np.float64)
        if a is not None:
            assert a.shape == (self.n_replicas, self.rl_n_actions), a.shape
        obs = rew = None
        if fetch:
            obs = np.empty((self.n_replicas, self.rl_n_obs), dtype=np.float32)
            rew = np.empty((self.n_replicas, self.rl_n_agents), dtype=np.float32)
        self._ck(self._lib.pedn_rl_step(self._h, None if a is None else a.ctypes.data_as(C.c_void_p), 0, int(t), int(action_gap),
                                        None if obs is None else obs.ctypes.data_as(C.c_void_p),
                                        None if rew is None else rew.ctypes.data_as(C.c_void_p)))
        return obs, rew

    def rl_step_device(self, actions_ptr, t, action_gap=1, ordered=False):
        """Same as rl_step with the action rows already resident in HBM (raw device pointer, e.g. torch ``data_ptr()``);
        observations and rewards stay in the device buffers (``rl_device_ptr``).  ordered: every launch on ``stream_ptr()`` (for a
        caller that chains the call to its own streams with events)."""
        self._ck(self._lib.pedn_rl_step(self._h, C.c_void_p(int(actions_ptr)), 2 if ordered else 1, int(t), int(action_gap), None, None))

    def rl_clock_begin(self, t):
        """The device-resident step clock set to step t (include/pedn.h: pedn_rl_clock_begin); the fractions of t must be prepared."""
        self._ck(self._lib.pedn_rl_clock_begin(self._h, int(t)))

    def rl_step_clocked(self, actions_ptr, action_gap=1, stream_ptr=0):
        """One env step with constant launch arguments on ``stream_ptr`` (0: the engine's stream); safe under stream capture."""
        self._ck(self._lib.pedn_rl_step_clocked(self._h, C.c_void_p(int(actions_ptr)) if actions_ptr else None, int(action_gap),
                                                C.c_void_p(int(stream_ptr)) if stream_ptr else None))

    def rl_clocked(self):
        return bool(self._lib.pedn_rl_clocked(self._h))

    def rl_clock_signature(self):
        """Hash of what the clocked step's launches carry by value: a captured graph is valid while it does not change."""
        return int(self._lib.pedn_rl_clock_signature(self._h))

    def rl_clock_end(self):
        """Ends the clocked section (synchronises the device); returns the next step to run."""
        t = C.c_int32(0)
        self._ck(self._lib.pedn_rl_clock_end(self._h, C.byref(t)))
        return int(t.value)

    def flush(self):
        """Everything pending enqueued on the engine's stream (pedn_flush): zero-copy consumers behind ``stream_ptr()`` see complete rows."""
        self._ck(self._lib.pedn_flush(self._h))

    def stream_ptr(self):
        """hipStream_t of the engine (``pedn_stream``), e.g. for ``torch.cuda.ExternalStream``."""
        return int(self._lib.pedn_stream(self._h) or 0)

    def rl_device_ptr(self, which):
        return self._lib.pedn_rl_device_ptr(self._h, int(which))

    def timer_begin(self):
        self._ck(self._lib.pedn_timer_begin(self._h))

    def timer_end(self):
        ms = C.c_float()
        self._ck(self._lib.pedn_timer_end(self._h, C.byref(ms)))
        return float(ms.value)

    # -- reads
    def _dtype(self, field):
        return np.float64 if field < 7 else np.float32

    def _ncols(self, field):
        return self.n_all if field < 4 else self.n_links

    def read_block(self, field, t0, t1, link0=0, link1=None, rep0=0, rep1=None):
        link1 = self._ncols(field) if link1 is None else link1
        rep1 = self.n_replicas if rep1 is None else rep1
        out = np.empty((t1 - t0, link1 - link0, rep1 - rep0), dtype=self._dtype(field))
        self._ck(self._lib.pedn_read(self._h, int(field), int(t0), int(t1), int(link0), int(link1), int(rep0), int(rep1),
                                     out.ctypes.data_as(C.c_void_p)))
        return out

    def history_rows(self, field):
        """Time indices the field keeps: T + 1, or the size of its ring in recent-history mode."""
        return self._ck(self._lib.pedn_history_rows(self._h, int(field)))

    def read_column(self, field, link, replica=0):
        return self.read_block(field, 0, self.T + 1, link, link + 1, replica, replica + 1).reshape(-1)

    def read_element(self, field, link, replica, t):
        return self.read_block(field, t, t + 1, link, link + 1, replica, replica + 1).reshape(-1)[0]

    def device_ptr(self, field):
        cols, stride = C.c_int64(), C.c_int64()
        p = self._lib.pedn_device_ptr(self._h, int(field), C.byref(cols), C.byref(stride))
        return p, cols.value, stride.value
