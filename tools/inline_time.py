"""The single-launch plan's variants side by side on one box: PEDN_INLINE_TF=0 (two launches per step), 1 (the slot waves compute
their own rows of turning fractions), 2 (helper waves compute them: sixteen waves per workgroup).  run(1, T) best of 4 and a Python
loop over network_loading(t), us per step; every field and flag compared with variant 0.

    python tools/inline_time.py [model:replicas ...]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from golden_util import ALL_FIELDS, DATA
from pednstream_amd import NetworkEnvGenerator
from pednstream_amd.network import LINK_FIELDS

cases = sys.argv[1:] or ["nine_intersections:256", "nine_intersections:1", "od_flow_example:1", "small_network:128", "45_intersections:64",
                         "45_intersections:512"]
for case in cases:
    name, R = case.split(":")
    R = int(R)
    out = {}
    for variant in ("0", "1", "2"):
        os.environ["PEDN_INLINE_TF"] = variant
        np.random.seed(7)
        net = NetworkEnvGenerator(DATA).create_network(name, verbose=False, n_replicas=R, rng_seed=11)
        e = net.engine()
        T = net.params["simulation_steps"]
        best = loop = 1e9
        for rep in range(4):
            net.reset()
            e.synchronize()
            t0 = time.perf_counter()
            net.run(1, T, check=False)
            e.synchronize()
            best = min(best, time.perf_counter() - t0)
        for rep in range(2):
            net.reset()
            e.synchronize()
            t0 = time.perf_counter()
            for t in range(1, T):
                net.network_loading(t)
            e.synchronize()
            loop = min(loop, time.perf_counter() - t0)
        rc, flags = e.error_flags()
        out[variant] = ({f: e.read_block(LINK_FIELDS[f][0], 0, T) for f in ALL_FIELDS}, best / (T - 1) * 1e6, loop / (T - 1) * 1e6, flags.copy())
        net.close()
    same = [all(np.array_equal(out["0"][0][f], out[k][0][f]) for f in ALL_FIELDS) and np.array_equal(out["0"][3], out[k][3]) for k in ("1", "2")]
    print(f"{case}: run() two launches {out['0'][1]:.2f}, own rows {out['1'][1]:.2f}, helper waves {out['2'][1]:.2f} us/step; "
          f"Python loop {out['0'][2]:.2f} / {out['1'][2]:.2f} / {out['2'][2]:.2f}; identical {same}", flush=True)
