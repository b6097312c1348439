"""Offline fuzz campaign (build container only): random small networks with random parameters through the REAL
reference (injected RNG) and through the C oracle; every one of the 13 per-link arrays and the turning fractions must
agree bit for bit.  Not a test (needs /root/reference); its result is quoted in DESIGN.md.

    python oracle/fuzz_vs_reference.py [n_cases] [first_seed]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness as rh  # noqa: E402

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_driver as od  # noqa: E402
from pednstream_amd import Network  # noqa: E402
from pednstream_amd.flatten import flatten_network  # noqa: E402


from fuzz_cases import random_case  # noqa: E402  (tests/fuzz_cases.py: shared with the GPU fuzz test)


SHORT = os.environ.get("PEDN_FUZZ_SHORT_LINKS", "0") != "0"      # corridors shorter than half a time step (zero-step look-backs)


def run(seed):
    adj, params, origins, dests = random_case(seed, short_links=SHORT)
    ref = rh.load_reference()
    import copy

    np.random.seed(seed)
    try:
        rnet = ref["network"].Network(adj.copy(), copy.deepcopy(params), origin_nodes=list(origins), destination_nodes=list(dests))
    except KeyError:
        return "skip (controller on no path: reference KeyError)"
    rnet, st, state, ex = rh.run_reference(None, seed=seed, replica=seed % 7, record_tf=True, network=rnet)
    np.random.seed(seed)
    mine = Network(adj.copy(), copy.deepcopy(params), origin_nodes=list(origins), destination_nodes=list(dests), verbose=False)
    for nid, node in rnet.nodes.items():
        if node.demand is not None:
            mine.nodes[nid].demand = np.asarray(node.demand, dtype=np.float64)
    model = flatten_network(mine)
    if model["max_degree"] > 8:
        return "skip (degree > 8)"
    o = od.Oracle(model, seed=seed, replica=seed % 7)
    T = params["simulation_steps"]
    tfh = []
    for t in range(1, T):
        o.step(t)
        tfh.append(o.tf())
    if o.flags():
        return f"skip (flags {o.flags()}: reference order-dependent / raises)"
    L = model["n_links"]
    for f in od.ALL_FIELDS:
        a, b = o.field(f)[:L, :T], state[f][:, :T]
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            return f"MISMATCH {f} first t={bad[:, 1].min()} n={len(bad)}"
    rtf = np.concatenate([ex["tf_hist"][int(nid)] for nid in rnet.nodes.keys()], axis=1)
    if not np.array_equal(np.array(tfh), rtf):
        return "MISMATCH turning fractions"
    return "ok"


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    tally = {}
    for seed in range(first, first + n):
        try:
            res = run(seed)
        except (ValueError, Warning, IndexError) as e:      # the reference itself raised (negative flow etc.)
            res = f"skip (reference raised {type(e).__name__})"
            if SHORT:
                print(seed, res, str(e)[:90], flush=True)
        key = res.split(" ")[0] if res.startswith("skip") or res == "ok" else res
        tally[res if res.startswith("MISMATCH") else key] = tally.get(res if res.startswith("MISMATCH") else key, 0) + 1
        if res.startswith("MISMATCH"):
            print(seed, res, flush=True)
    print(tally)
