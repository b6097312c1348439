"""The launch plans against each other under random sequences of calls: engine A steps under the plan the engine picks (owner-wave link
update with its pending state, single-launch plan of small batches, two chains for the larger batch), engine B under the plain plan
(two launches per step on one stream, PEDN_LINK_OWNER=0 PEDN_INLINE_TF=0 PEDN_STREAMS=1).  Ranges of steps, single steps, repeated
steps, reads of random fields, width / turning-fraction / demand setters, resets (lazy on A), in random order: every row of every field,
the turning fractions and the flags identical whenever compared.

    python tools/gpu_fuzz_plans.py 950000 950300
"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from golden_util import ALL_FIELDS
from fuzz_cases import random_case
from pednstream_amd import Network
from pednstream_amd.flatten import flatten_network
from pednstream_amd.network import LINK_FIELDS

PLAIN = {"PEDN_LINK_OWNER": "0", "PEDN_INLINE_TF": "0", "PEDN_STREAMS": "1"}
lo, hi = int(sys.argv[1]), int(sys.argv[2])
ran = skipped = ops_total = n_flagged = 0
plans = {}
for seed in range(lo, hi):
    if seed > lo and (seed - lo) % 100 == 0:
        print(f"#   ... seed {seed} of {lo}..{hi}: {ran} networks so far", flush=True)
    adj, params, origins, dests = random_case(seed)
    R = (384 if seed % 8 == 0 else 256) if seed % 4 == 0 else 3   # every fourth network with a batch that steps as two chains (384: 256 + 128)
    nets = []
    try:
        for plain in (False, True):
            keep = {k: os.environ.get(k) for k in PLAIN}
            if plain:
                os.environ.update(PLAIN)
            elif R >= 256:
                os.environ["PEDN_STREAMS"] = "2"
                os.environ["PEDN_STREAM_PROBE"] = "0"
            np.random.seed(seed)
            net = Network(adj, copy.deepcopy(params), origin_nodes=origins, destination_nodes=dests, verbose=False, n_replicas=R, rng_seed=seed,
                          history="recent" if seed % 5 == 0 else "full")
            net.engine()                                 # the plan is read from the environment when the engine is created
            for k, v in keep.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
            os.environ.pop("PEDN_STREAM_PROBE", None)
            nets.append(net)
    except KeyError:
        skipped += 1
        continue
    if flatten_network(nets[0])["max_degree"] > 8:
        skipped += 1
        continue
    info = nets[0].engine().plan_info()
    plans[(info["chains"], info["link_update_by_next_node_kernel"])] = plans.get((info["chains"], info["link_update_by_next_node_kernel"]), 0) + 1
    T = params["simulation_steps"]
    rng = np.random.default_rng(seed)
    recent = seed % 5 == 0
    t = 1
    log = []
    links = list(nets[0].links)
    dyn_nodes = [n for n in nets[0].nodes.values() if len(n.incoming_links) > 2]

    def flagged():
        fa, fb = (n.engine().error_flags()[1] for n in nets)
        assert np.array_equal(fa, fb), (seed, "flags", log)
        return bool((fa & 16).any())                     # a zero look-back: order-dependent by definition

    def compare():
        if flagged():
            return False
        last = t - 1
        first = max(0, last - 1) if recent else 0        # recent-history mode: most fields are short rings
        for f in ALL_FIELDS:
            hi_t = last if (recent and f in ("sending_flow", "receiving_flow")) else last + 1
            if hi_t <= first:
                continue
            a, b = (n.engine().read_block(LINK_FIELDS[f][0], first, hi_t) for n in nets)
            if not np.array_equal(a, b):
                d = np.argwhere(a != b)
                raise AssertionError((seed, f, "first differing (t, column, replica)", d[0].tolist(), "of", len(d), log))
        for nd in nets[0].nodes.values():
            for r in (0, R - 1):
                ta, tb = (n.engine().get_turning_fractions(nd.index, r) for n in nets)
                assert np.array_equal(ta, tb), (seed, "tf", nd.index, r, log)
        return True

    stopped = False
    for op in range(int(rng.integers(8, 22))):
        kind = rng.choice(["run", "step", "steps", "repeat", "read", "width", "tf", "demand", "reset", "compare"],
                          p=[0.22, 0.14, 0.14, 0.04, 0.1, 0.08, 0.06, 0.06, 0.08, 0.08])
        ops_total += 1
        log.append((str(kind), t))
        if kind == "run" and t < T:
            n = min(int(rng.integers(1, 40)), T - t)
            for net in nets:
                net.run(t, t + n, check=False)
            t += n
        elif kind == "step" and t < T:
            for net in nets:
                net.network_loading(t)
            t += 1
        elif kind == "steps" and t < T:   # the reference's loop: call after call (from the third one on the chains stay forked across the calls)
            n = min(int(rng.integers(3, 14)), T - t)
            for net in nets:
                for k in range(n):
                    net.network_loading(t + k)
            t += n
        elif kind == "repeat" and t > 3 and not recent:
            for net in nets:
                net.engine().step(t - 1)
        elif kind == "read" and t > 1:
            f = ALL_FIELDS[int(rng.integers(0, len(ALL_FIELDS)))]
            tt = t - 1 if not (f in ("sending_flow", "receiving_flow")) else max(t - 2, 0)
            a, b = (n.engine().read_block(LINK_FIELDS[f][0], tt, tt + 1) for n in nets)
            if not flagged():
                assert np.array_equal(a, b), (seed, f, tt, "read", log)
        elif kind == "width":
            l = links[int(rng.integers(0, len(links)))]
            w = float(rng.uniform(0.3, 2.0))
            for net in nets:
                net.links[l].back_gate_width = w
        elif kind == "tf" and dyn_nodes:
            nd = dyn_nodes[int(rng.integers(0, len(dyn_nodes)))]
            n_tf = len(np.asarray(nd.turning_fractions))
            m = int(round((1 + np.sqrt(1 + 4 * n_tf)) / 2))          # m slots, m (m - 1) fractions
            if m < 2 or m * (m - 1) != n_tf:
                continue
            tf = rng.dirichlet(np.ones(m - 1), size=m).reshape(-1)
            for net in nets:
                net.update_turning_fractions_per_node([nd.node_id], [tf])
        elif kind == "demand" and origins:
            nid = origins[int(rng.integers(0, len(origins)))]
            rows = rng.poisson(rng.uniform(2, 30), (R, T)).astype(np.float64)
            for net in nets:
                net.set_demand_matrix(nid, rows)
        elif kind == "reset":
            nets[0].reset(lazy=True)
            nets[1].reset()
            t = 1
        elif kind == "compare" and t > 1:
            if not compare():
                stopped = True
                break
        if os.environ.get("FUZZ_COMPARE_EACH") and t > 1 and not compare():
            stopped = True
            break
    if not stopped and t > 1:
        stopped = not compare()
    n_flagged += stopped
    for net in nets:
        net.close()
    ran += 1
print(f"the engine's own launch plans == two launches per step on one stream: {ran} random networks (3, 256 or 384 replicas, every fifth in recent-history mode), "
      f"{ops_total} random calls; plans of engine A (chains, link update by the next node kernel): {plans}; every field, turning fraction and flag identical "
      f"whenever compared; {n_flagged} scenarios stopped at a zero look-back, {skipped} networks skipped")
