"""When do the workgroups of link_turn_kernel start and end?  (profiling build, `make -C pednstream_amd/csrc phase-profile`)

One launch of the fused kernel (turning fractions of t+1 | link update of t | observations) on a network x replicas, one chain:
per role the number of workgroups and the distribution of their start and end times (us after the first workgroup started),
and how many workgroups were resident over time.

    python tools/lt_timeline.py delft 1024            # PEDN_TF_HEAVY_GROUPS selects the order
    python tools/lt_timeline.py 45_intersections 2048 rl   # the launch behind node_kernel of the batched RL step (with observations)
"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "pednstream_amd", "csrc", "libpedn_hip_phase.so")
os.environ["PEDN_HIP_LIB"] = LIB
os.environ.setdefault("PEDN_STREAMS", "1")

from bench import replica_demand  # noqa: E402
from pednstream_amd import NetworkEnvGenerator  # noqa: E402

ROLES = {0: "turning fractions, long rows", 1: "link update", 2: "turning fractions, short rows", 3: "observations"}


def main():
    lib = ctypes.CDLL(LIB)
    network = sys.argv[1] if len(sys.argv) > 1 else "delft"
    R = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
    rl = len(sys.argv) > 3 and sys.argv[3] == "rl"
    if rl:
        import time

        from pednstream_amd.rl_env import VecPedNetEnv
        env = VecPedNetEnv(network, n_envs=R, obs_mode="option3", action_gap=1, seed=0, data_dir=os.path.join(ROOT, "data"))
        net = env.network
        e = net.engine()
        acts = np.random.default_rng(0).uniform(0.0, 1.0, (R, env.n_actions)) * env.action_high
        for t in range(1, 200):
            e.rl_step(acts, t, 1, fetch=False)
        e.synchronize()
    else:
        net = NetworkEnvGenerator(os.path.join(ROOT, "data")).create_network(network, verbose=False, n_replicas=R, rng_seed=0)
        e = net.engine()
        for nid in net.origin_nodes:
            e.set_demand_matrix(net.nodes[nid].index, np.stack([replica_demand(net.simulation_steps, r) for r in range(R)]))
        e.run(1, 200)
        e.synchronize()
    N = 1 << 15
    out = (ctypes.c_ulonglong * (N * 4))()
    print(f"== {network} x {R}, one chain, PEDN_TF_HEAVY_GROUPS={os.environ.get('PEDN_TF_HEAVY_GROUPS', 'default')}: link_turn_kernel of steps 200..203")
    for t in range(200, 204):
        lib.pedn_debug_lt_timeline(None, 0, 1)
        if rl:
            t0w = time.perf_counter()
            e.rl_step(acts, t, 1, fetch=False)
            e.synchronize()
            ms = [0.0, 0.0, (time.perf_counter() - t0w) * 1e3]     # wall time of the whole env step, not the launch
        else:
            ms = e.profile_step(t)
        lib.pedn_debug_lt_timeline(out, N, 0)
        a = np.array(out[:], dtype=np.float64).reshape(N, 4)
        a = a[a[:, 1] > 0]
        t0 = a[:, 1].min()
        start, end = (a[:, 1] - t0) / 100.0, (a[:, 2] - t0) / 100.0       # 100 MHz -> us
        print(f"-- step {t}: {len(a)} workgroups, {'env step (wall, synchronised)' if rl else 'launch'} {ms[2] * 1e3:.1f} us{'' if rl else ' by its dispatch timestamps'}, last wave ends at {end.max():.1f} us")
        print("   role                           wgs |  start: min  median     max |  end: min  median     max | lifetime: median  max")
        for role, name in ROLES.items():
            m = a[:, 0] == role
            if m.any():
                s_, e_ = start[m], end[m]
                print(f"   {name:28s} {m.sum():5d} | {s_.min():10.1f} {np.median(s_):7.1f} {s_.max():7.1f} | {e_.min():8.1f} {np.median(e_):7.1f} {e_.max():7.1f} |"
                      f" {np.median(e_ - s_):16.1f} {np.max(e_ - s_):4.1f}")
        grid = np.arange(0.0, end.max() + 1.0, 1.0)
        res = [(int(((start <= x) & (end > x) & (a[:, 0] == r)).sum())) for x in grid for r in ROLES]
        res = np.array(res).reshape(len(grid), len(ROLES))
        print("   resident workgroups by role at t = 0, 1, 2 ... us  (long | link | short | obs)")
        print("   " + "  ".join(f"{int(x)}:{'/'.join(str(v) for v in row)}" for x, row in zip(grid, res)))
    (env if rl else net).close()


if __name__ == "__main__":
    main()
