"""Test-side driver of the CPU oracle (oracle/libpedn_oracle.so).  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from pednstream_amd.engine import ModelDesc, build_model_desc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libpedn_oracle.so")

F64_FIELDS = ["inflow", "outflow", "cumulative_inflow", "cumulative_outflow", "sending_flow", "receiving_flow",
              "back_gate_width_data"]
F32_FIELDS = ["travel_time", "avg_travel_time", "num_pedestrians", "density", "speed", "link_flow"]
ALL_FIELDS = F64_FIELDS + F32_FIELDS

_lib = None


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libpedn_oracle.so"])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build_oracle()
        L = C.CDLL(LIB)
        P = C.c_void_p
        L.pedn_oracle_create.restype = P
        L.pedn_oracle_create.argtypes = [C.POINTER(ModelDesc), C.c_uint64, C.c_int32, C.c_int32]
        L.pedn_oracle_destroy.argtypes = [P]
        L.pedn_oracle_reset.argtypes = [P]
        L.pedn_oracle_reseed.argtypes = [P, C.c_uint64, C.c_int32]
        L.pedn_oracle_step.argtypes = [P, C.c_int]
        L.pedn_oracle_run.argtypes = [P, C.c_int, C.c_int]
        L.pedn_oracle_run_many.argtypes = [C.POINTER(P), C.c_int, C.c_int, C.c_int]
        L.pedn_oracle_set_demand.argtypes = [P, C.c_int, C.POINTER(C.c_double), C.c_int]
        L.pedn_oracle_set_od_weights.argtypes = [P, C.c_int, C.POINTER(C.c_double), C.c_int]
        L.pedn_oracle_set_width.argtypes = [P, C.c_int, C.c_int, C.c_double]
        L.pedn_oracle_set_tf.argtypes = [P, C.c_int, C.POINTER(C.c_double), C.c_int]
        L.pedn_oracle_tf.restype = C.POINTER(C.c_double)
        L.pedn_oracle_tf.argtypes = [P]
        L.pedn_oracle_field.restype = C.c_void_p
        L.pedn_oracle_field.argtypes = [P, C.c_int]
        L.pedn_oracle_flags.restype = C.c_uint32
        L.pedn_oracle_flags.argtypes = [P]
        L.pedn_oracle_tally.restype = None
        L.pedn_oracle_tally.argtypes = [P, C.POINTER(C.c_uint64)]
        L.pedn_oracle_powf.restype = C.c_float
        L.pedn_oracle_powf.argtypes = [C.c_float, C.c_float]
        L.pedn_oracle_exp.restype = C.c_double
        L.pedn_oracle_exp.argtypes = [C.c_double]
        L.pedn_oracle_philox.argtypes = [C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32]
        L.pedn_oracle_binomial.restype = C.c_int64
        L.pedn_oracle_binomial.argtypes = [C.c_int64, C.c_double, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.pedn_oracle_normal.restype = C.c_double
        L.pedn_oracle_normal.argtypes = [C.c_double, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        _lib = L
    return _lib


class Oracle:
    """One replica of the CPU restatement."""

    def __init__(self, model: dict, seed=0, replica=0, mode="philox"):
        self.L = lib()
        self.model = model
        desc, self._keep = build_model_desc(model)
        self.h = C.c_void_p(self.L.pedn_oracle_create(C.byref(desc), int(seed), int(replica),
                                                      {"philox": 0, "meanfield": 1}[mode]))
        self.T1 = int(model["T"]) + 1
        self.n_links = int(model["n_links"])
        self.n_all = self.n_links + int(model["n_vlinks"])

    def close(self):
        if self.h:
            self.L.pedn_oracle_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def step(self, t):
        return self.L.pedn_oracle_step(self.h, int(t))

    def run(self, t0, t1):
        return self.L.pedn_oracle_run(self.h, int(t0), int(t1))

    def reset(self, seed=None, replica=None):
        self.L.pedn_oracle_reset(self.h)
        if seed is not None:
            self.L.pedn_oracle_reseed(self.h, int(seed), int(replica or 0))

    def flags(self):
        return int(self.L.pedn_oracle_flags(self.h))

    def tally(self):
        """Paths of cal_sending_flow taken since the oracle was created (resets keep counting): calls, past the free-flow gate,
        positive before the draw, diffusion look-backs read, activity draws."""
        out = (C.c_uint64 * 5)()
        self.L.pedn_oracle_tally(self.h, out)
        return [int(x) for x in out]

    def set_demand(self, node_index, values):
        v = np.ascontiguousarray(values, dtype=np.float64)
        self.L.pedn_oracle_set_demand(self.h, int(node_index), v.ctypes.data_as(C.POINTER(C.c_double)), len(v))

    def set_width(self, which, link, value):
        self.L.pedn_oracle_set_width(self.h, int(which), int(link), float(value))

    def set_tf(self, node_index, tf):
        v = np.ascontiguousarray(tf, dtype=np.float64)
        self.L.pedn_oracle_set_tf(self.h, int(node_index), v.ctypes.data_as(C.POINTER(C.c_double)), len(v))

    def tf(self):
        n = int(self.model["n_turns"])
        return np.ctypeslib.as_array(self.L.pedn_oracle_tf(self.h), shape=(n,)).copy()

    def field(self, name):
        """[columns, T+1] copy of one history field."""
        fid = ALL_FIELDS.index(name)
        cols = self.n_all if fid < 4 else self.n_links
        dt = np.float64 if fid < 7 else np.float32
        ptr = self.L.pedn_oracle_field(self.h, fid)
        buf = (C.c_char * (cols * self.T1 * np.dtype(dt).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=dt).reshape(cols, self.T1).copy()


def run_many(oracles, t0, t1):
    L = lib()
    arr = (C.c_void_p * len(oracles))(*[o.h for o in oracles])
    return L.pedn_oracle_run_many(arr, len(oracles), int(t0), int(t1))
