"""Where the lifetime of a node_kernel wave goes: s_memtime stamps at the phase boundaries, summed over all waves.

Needs the profiling build of the engine (`make -C pednstream_amd/csrc phase-profile`, adds ~10 % to the kernel) and a GPU:

    python tools/phase_profile.py melbourne delft 45_intersections:2048 nine_intersections:256      # network[:replicas], default 1024
"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "pednstream_amd", "csrc", "libpedn_hip_phase.so")
os.environ["PEDN_HIP_LIB"] = LIB
# the ten 64-bit stamps live in registers: at node_kernel's normal budget (8 waves per SIMD, 64 VGPRs) the instrumented build spills
# 25..32 vector registers and the picture is the spills'; at 6 waves (80 VGPRs) it spills none

from bench import replica_demand  # noqa: E402
from pednstream_amd import NetworkEnvGenerator  # noqa: E402

PHASES = ["kernel entry -> slot record", "slot record -> batch of history loads", "sending flow (look-backs, diffusion, binomial)",
          "receiving flow", "turning-fraction row, P*s into LDS", "barrier 1 (slowest wave of the block)", "column pass",
          "barrier 2", "row sums + stores"]


def main():
    lib = ctypes.CDLL(LIB)
    for spec in sys.argv[1:] or ["melbourne"]:
        network, _, reps = spec.partition(":")
        gen = NetworkEnvGenerator(os.path.join(ROOT, "data"))
        R = int(reps) if reps else 1024
        net = gen.create_network(network, verbose=False, n_replicas=R, rng_seed=0)
        e = net.engine()
        for nid in net.origin_nodes:
            e.set_demand_matrix(net.nodes[nid].index, np.stack([replica_demand(net.simulation_steps, r) for r in range(R)]))
        e.run(1, 150)
        e.synchronize()
        lib.pedn_debug_phases(None, 1)
        t0 = time.perf_counter()
        e.run(150, 250)
        e.synchronize()
        wall = (time.perf_counter() - t0) / 100
        out = (ctypes.c_ulonglong * 16)()
        lib.pedn_debug_phases(out, 0)
        o = np.array(out[:], dtype=np.float64)
        n, total = o[10], o[11]
        print(f"== {network} x {R}: {int(n / 100)} active waves per launch, mean wave lifetime {total / n:.0f} s_memtime ticks "
              f"(instrumented step, built for 6 waves per SIMD: {wall * 1e6:.1f} us)")
        for i, name in enumerate(PHASES, start=1):
            print(f"   {name:48s} {o[i] / n:9.0f} ticks  {100 * o[i] / total:5.1f} %")
        net.close()


if __name__ == "__main__":
    main()
