"""Lazy reset (pedn_reset_lazy) against the ordinary reset: two engines of the same random network execute the same random sequence of
calls -- ranges of steps, single steps, steps that jump ahead or repeat an earlier one, reads, demand and width changes, resets (one
engine lazily, the other by clearing everything) -- and must agree in every row of every field, whenever they are compared.

    python tools/gpu_fuzz_lazy.py 900000 900300
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

lo, hi = int(sys.argv[1]), int(sys.argv[2])
ran = skipped = ops_total = n_flagged = 0
for seed in range(lo, hi):
    if seed > lo and (seed - lo) % 100 == 0:
        print(f"#   ... seed {seed} of {lo}..{hi}: {ran} networks so far", flush=True)
    adj, params, origins, dests = random_case(seed)
    nets = []
    try:
        for _ in range(2):
            np.random.seed(seed)
            nets.append(Network(adj, copy.deepcopy(params), origin_nodes=origins, destination_nodes=dests, verbose=False, n_replicas=3, rng_seed=seed))
    except KeyError:
        skipped += 1
        continue
    if flatten_network(nets[0])["max_degree"] > 8:
        skipped += 1
        continue
    T = params["simulation_steps"]
    rng = np.random.default_rng(seed)
    t = 1                                    # next consecutive step
    log = []
    link0 = next(iter(nets[0].links))

    def flagged():
        """A raise site of the reference was hit (typically PEDN_F_SAME_STEP: a step that jumps into rows whose avg_travel_time is 0
        looks back zero steps, i.e. reads the row other waves are writing -- the reference's result depends on its node order there,
        and so does a kernel's): the flags must agree, the numbers behind them are not compared."""
        fa, fb = (n.engine().error_flags()[1] for n in nets)
        assert np.array_equal(fa != 0, fb != 0), (seed, "flags", log)
        return bool(fa.any())

    def compare(t1):
        for f in ALL_FIELDS:
            a, b = (n.engine().read_block(LINK_FIELDS[f][0], 0, t1) for n in nets)
            if not np.array_equal(a, b):
                d = np.argwhere(a != b)
                raise AssertionError((seed, f, t1, "first differing (t, column, replica)", d[0].tolist(), "of", len(d), d[:6].tolist(), float(a[tuple(d[0])]), float(b[tuple(d[0])]), log))
        fa, fb = (n.engine().error_flags()[1] for n in nets)
        assert np.array_equal(fa, fb), (seed, "flags")

    for op in range(int(rng.integers(8, 20))):
        kind = rng.choice(["run", "step", "jump", "repeat", "read", "demand", "width", "reset", "compare"], p=[0.25, 0.15, 0.1, 0.05, 0.1, 0.08, 0.07, 0.12, 0.08])
        ops_total += 1
        log.append((str(kind), t))
        if kind == "run" and t < T:
            n = min(int(rng.integers(1, 30)), T - t)
            for net in nets:
                net.run(t, t + n, check=False)
            t += n
        elif kind == "step" and t < T:
            for net in nets:
                net.engine().step(t)
            t += 1
        elif kind == "jump" and t + 3 < T:
            t = min(t + int(rng.integers(2, 12)), T - 1)
            W = int(round(100.0 / params["unit_time"]))
            if rng.random() < 0.6 and t - 1 >= W:          # most jumps stay inside the window (avg_travel_time = travel_time[0] there:
                t = max(2, min(t, W - 1))                  # no zero look-back), so that the scenario goes on
            log.append(("to", t))
            for net in nets:
                net.engine().step(t)
            t += 1
        elif kind == "repeat" and t > 2:
            back = int(rng.integers(1, t - 1))
            log.append(("back", back))
            for net in nets:
                net.engine().step(back)
        elif kind == "read":
            if flagged():
                break
            f = ALL_FIELDS[int(rng.integers(0, len(ALL_FIELDS)))]
            t1 = int(rng.integers(1, T + 2))
            a, b = (n.engine().read_block(LINK_FIELDS[f][0], 0, t1) for n in nets)
            assert np.array_equal(a, b), (seed, f, t1, "read")
        elif kind == "demand" and origins:
            nid = origins[int(rng.integers(0, len(origins)))]
            rows = rng.poisson(rng.uniform(2, 30), (3, T)).astype(np.float64)
            for net in nets:
                net.set_demand_matrix(nid, rows)
        elif kind == "width":
            w = float(rng.uniform(0.3, 2.0))
            for net in nets:
                net.links[link0].back_gate_width = w
        elif kind == "reset":
            nets[0].reset(lazy=True)
            nets[1].reset()
            t = 1
        elif kind == "compare":
            if flagged():
                break
            compare(T + 1)
        if kind in ("jump", "repeat") and flagged():
            break
    stopped = flagged()
    n_flagged += stopped
    if not stopped:
        compare(T + 1)
    for net in nets:
        net.close()
    ran += 1
print(f"lazy reset == ordinary reset: {ran} random networks x 3 replicas, {ops_total} random calls (ranges, single / jumping / repeated steps, reads, "
      f"demand and gate changes, resets), every row of every field and the flags identical whenever compared; {n_flagged} scenarios stopped at a raise site "
      f"(same flags on both engines), {skipped} networks skipped")
