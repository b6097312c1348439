"""The persistent plan of pedn_run (ranges of small networks as ONE launch, node_persist_kernel) against a launch per step: every
field bit for bit, and the time of run(1, T) under both.

    python tools/persist_time.py [model:replicas ...]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from golden_util import ALL_FIELDS, DATA
from pednstream_amd import NetworkEnvGenerator
from pednstream_amd.network import LINK_FIELDS

cases = sys.argv[1:] or ["nine_intersections:256", "nine_intersections:64", "nine_intersections:1", "six_node:1", "od_flow_example:1",
                         "long_corridor:64", "melbourne:1", "melbourne:64", "45_intersections:64", "delft:1", "small_network:128"]
for case in cases:
    name, R = case.split(":")
    R = int(R)
    out = {}
    for persist in ("0", "1"):
        os.environ["PEDN_PERSIST"] = persist
        np.random.seed(7)
        try:
            net = NetworkEnvGenerator(DATA).create_network(name, verbose=False, n_replicas=R, rng_seed=11)
        except Exception as e:
            print(f"{case}: {type(e).__name__}: {e}")
            break
        e = net.engine()
        T = net.params["simulation_steps"]
        info = e.plan_info()
        best = 1e9
        for rep in range(4):
            net.reset()
            e.synchronize()
            t0 = time.perf_counter()
            net.run(1, T, check=False)
            e.synchronize()
            best = min(best, time.perf_counter() - t0)
        # a second shape of calls: short ranges, a single step in between, a lazy reset before
        net.reset(lazy=True)
        net.run(1, 4, check=False); net.network_loading(4); net.run(5, 9, check=False); net.run(9, T - 2, check=False); net.run(T - 2, T, check=False)
        rc, flags = e.error_flags()
        out[persist] = ({f: e.read_block(LINK_FIELDS[f][0], 0, T) for f in ALL_FIELDS}, best / (T - 1) * 1e6, info, flags.copy())
        net.close()
    else:
        same = all(np.array_equal(out["0"][0][f], out["1"][0][f]) for f in ALL_FIELDS) and np.array_equal(out["0"][3], out["1"][3])
        print(f"{case}: launch per step {out['0'][1]:.2f} us/step, persistent ({out['1'][2]['persistent_ranges']}) {out['1'][1]:.2f} us/step, "
              f"identical {same}, flags {int(out['1'][3].max())}", flush=True)
