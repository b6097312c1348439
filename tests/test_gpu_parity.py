"""GPU parity tests (run on the MI355X box: pytest -m gpu).  Everything goes through the C-ABI of
pednstream_amd/csrc/libpedn_hip.so; the CPU oracle and the reference goldens are the checkers."""
import ctypes as C

import numpy as np
import pytest

import oracle_driver as od
from golden_util import ALL_FIELDS, DIGEST_CASES, Golden, apply_mutation, build_network, compare_digests, compare_fields
from pednstream_amd import engine as eng
from pednstream_amd.flatten import flatten_network
from pednstream_amd.network import LINK_FIELDS

pytestmark = pytest.mark.gpu

CASES = ["six_node_full", "nine_full", "long_corridor_full", "small_network_full", "i45_prefix", "delft_prefix",
         "melbourne_prefix", "nine_meanfield", "six_node_gate", "forky", "odd_params", "odd_separators", "star8", "edge_window_gt_T", "edge_empty",
         "butterfly_scA_full", "butterfly_scB_full", "butterfly_scC_full", "one_intersection_full", "two_coordinators_prefix", "spike_callable", "melbourne_callable", "forky_front", "edge_T2", "edge_T3"]


def _dev_math(op, a, b=None, seed=0):
    a = np.ascontiguousarray(a, dtype=np.float64)
    out = np.empty_like(a)
    P = C.POINTER(C.c_double)
    bp = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
    rc = eng.lib().pedn_device_math(0, op, len(a), a.ctypes.data_as(P), None if bp is None else bp.ctypes.data_as(P),
                                    seed, out.ctypes.data_as(P))
    assert rc == 0, eng.lib().pedn_last_error(None)
    return out


def test_device_arithmetic_matches_oracle_bit_for_bit():
    L = od.lib()
    rng = np.random.default_rng(3)
    x = np.concatenate([rng.random(50000).astype(np.float32).astype(np.float64), [0.0, 1.0, 0.5, 2.0 ** -24, 1e-30]])
    for y in (0.8, 2.0, 3.0):
        yy = np.full_like(x, np.float32(y))
        got = _dev_math(0, x, yy)
        want = np.array([L.pedn_oracle_powf(float(v), float(np.float32(y))) for v in x], dtype=np.float64)
        assert np.array_equal(got, want), f"powf y={y}"
    xs = np.concatenate([np.linspace(-60, 5, 60001), -rng.random(20000) * 30])
    assert np.array_equal(_dev_math(1, xs), np.array([L.pedn_oracle_exp(float(v)) for v in xs]))
    pos = rng.random(50000) * 1e4
    assert np.array_equal(_dev_math(2, pos), np.sqrt(pos))
    den = rng.random(50000) + 1e-3
    assert np.array_equal(_dev_math(3, pos, den), pos / den)
    f32 = (pos.astype(np.float32) / den.astype(np.float32)).astype(np.float64)
    assert np.array_equal(_dev_math(4, pos.astype(np.float32).astype(np.float64), den.astype(np.float32).astype(np.float64)), f32)
    n = rng.choice([0, 1, 3, 16, 17, 100, 5000, 40000], size=4000).astype(np.float64)
    p = rng.random(4000).astype(np.float32).astype(np.float64)
    got = _dev_math(5, n, p, seed=12345678901)
    want = np.array([L.pedn_oracle_binomial(int(a), float(b), 12345678901, i, 7, 11, 0) for i, (a, b) in enumerate(zip(n, p))], dtype=np.float64)
    assert np.array_equal(got, want)
    sig = np.full(4000, 0.05)
    got = _dev_math(6, sig, seed=77)
    want = np.array([L.pedn_oracle_normal(0.05, 77, i, 7, 11) for i in range(4000)])
    assert np.array_equal(got, want)


def _run_engine_on_golden(g, n_replicas=1):
    net = build_network(g, n_replicas=n_replicas, replica_offset=g.replica, rng_seed=g.seed, rng_mode=g.mode)
    last = g.steps
    tfh = []
    for t in range(0, last):
        if t > 0:
            net.network_loading(t)
            tfh.append(np.concatenate([net._engine.get_turning_fractions(nd.index, 0) for nd in net.nodes.values()]))
        for mut in g.mutations:          # t = 0: changes made before the first step (examples/forky_queues.py:112)
            if mut[0] == t:
                apply_mutation(net, mut)
    return net, np.array(tfh)


@pytest.mark.parametrize("case", CASES)
def test_engine_reproduces_reference_goldens(case):
    """HIP path vs the reference itself (goldens captured under the injected RNG): bit-exact on all 13 arrays."""
    g = Golden(case)
    net, tfh = _run_engine_on_golden(g)
    e = net._engine
    rc, flags = e.error_flags()
    assert rc == 0
    L = e.n_links

    def field(name):
        fid = LINK_FIELDS[name][0]
        return e.read_block(fid, 0, g.steps)[:, :, 0].T      # [columns, t]

    problems = compare_fields(field, g, L, g.steps)
    assert not problems, "\n".join(problems)
    if e.n_all > L:
        for tag, off in (("vin", 0), ("vout", 1)):
            for name in ("inflow", "outflow", "cumulative_inflow", "cumulative_outflow"):
                mine = field(name)[L + off::2, :g.steps]
                assert np.array_equal(mine, g.state(f"{tag}_{name}")[:, :g.steps]), (tag, name)
    assert np.array_equal(tfh, g.z["tf_hist"])           # turning fractions bit-exact as well
    # link views (reference attribute surface) read the same numbers
    key = tuple(g.static("link_uv")[0])
    lk = net.links[key]
    assert np.array_equal(np.asarray(lk.cumulative_inflow)[:g.steps], g.state("cumulative_inflow")[0, :g.steps])
    assert lk.density[g.steps - 1] == g.state("density")[0, g.steps - 1]
    net.close()


def replica_demand(T, r, base=20.0, peak=25.0):
    t = np.arange(T)
    lam = base + peak * np.exp(-(t - T / 4) ** 2 / (2 * (T / 20) ** 2)) + peak * np.exp(-(t - 3 * T / 4) ** 2 / (2 * (T / 20) ** 2))
    return np.random.default_rng(1000 + r).poisson(lam).astype(np.float64)


@pytest.mark.parametrize("case", DIGEST_CASES)
def test_engine_reproduces_full_horizon_reference_runs(case):
    """The two headline networks over all 499 steps against the reference itself (per-step digests over all links + complete
    arrays of 16 links + turning-fraction digests); melbourne also under heavy demand."""
    g = Golden(case)
    net, tfh = _run_engine_on_golden(g)
    e = net._engine
    rc, flags = e.error_flags()
    assert rc == 0
    problems = compare_digests(lambda name: e.read_block(LINK_FIELDS[name][0], 0, g.steps)[:, :, 0].T, g, e.n_links, tfh)
    assert not problems, "\n".join(problems)
    net.close()


def test_config2_nine_intersections_256_replicas_vs_oracle_and_goldens():
    """BASELINE config #2: nine_intersections x 256 replicas, per-replica demand and RNG key; replicas 0..3 against the
    reference goldens, replicas 0..7 and 255 against the CPU oracle, all bit-exact (f32 fields included)."""
    R, steps = 256, 160
    g0 = Golden("nine_replica0")
    net = build_network(g0, n_replicas=R, rng_seed=0)
    e = net.engine()
    origins = {0: 0, 8: 1, 2: 2}
    for r in range(R):
        for nid, k in origins.items():
            e.set_demand(net.nodes[nid].index, replica_demand(500, 3 * r + k, peak=50.0 if nid == 2 else 25.0), replica=r)
    net.run(1, steps)
    blocks = {name: e.read_block(LINK_FIELDS[name][0], 0, steps) for name in ALL_FIELDS}    # [t, L, R]
    for r in range(4):
        g = Golden(f"nine_replica{r}")
        problems = compare_fields(lambda name: blocks[name][:, :, r].T, g, e.n_links, steps)
        assert not problems, f"replica {r}:\n" + "\n".join(problems)
    model = flatten_network(net)
    for r in (0, 1, 2, 3, 4, 5, 6, 7, 255):
        o = od.Oracle(model, seed=0, replica=r)
        for nid, k in origins.items():
            o.set_demand(net.nodes[nid].index, replica_demand(500, 3 * r + k, peak=50.0 if nid == 2 else 25.0))
        o.run(1, steps)
        for name in ALL_FIELDS:
            mine = blocks[name][:, :e.n_links, r].T
            assert np.array_equal(mine, o.field(name)[:e.n_links, :steps]), (r, name)
    net.close()


@pytest.mark.parametrize("name,replicas,steps", [("delft", 64, 60), ("melbourne", 128, 120), ("45_intersections", 64, 200),
                                                 ("delft", 192, 500), ("melbourne", 256, 500), ("45_intersections", 128, 700),
                                                 ("nine_intersections", 64, 500), ("small_network", 64, 500)])
def test_engine_equals_oracle_on_large_networks(name, replicas, steps):
    """Same seeded inputs through the HIP path and the CPU oracle; sampled replicas, all fields bit-exact."""
    from pednstream_amd import NetworkEnvGenerator
    from golden_util import DATA

    np.random.seed(99)
    net = NetworkEnvGenerator(DATA).create_network(name, verbose=False, n_replicas=replicas, rng_seed=5, replica_offset=1000)
    net.run(1, steps)
    e = net._engine
    model = flatten_network(net)
    for r in (0, replicas // 2, replicas - 1):
        o = od.Oracle(model, seed=5, replica=1000 + r)
        o.run(1, steps)
        assert o.flags() == 0
        for fname in ALL_FIELDS:
            fid = LINK_FIELDS[fname][0]
            mine = e.read_block(fid, 0, steps, rep0=r, rep1=r + 1)[:, :, 0].T
            assert np.array_equal(mine[:e.n_links], o.field(fname)[:e.n_links, :steps]), (r, fname)
    net.close()


@pytest.mark.parametrize("name,steps", [("nine_intersections", 120), ("delft", 40), ("long_corridor", 150)])
def test_launch_variants_give_identical_histories(name, steps, monkeypatch):
    """The engine picks its launch plan per model: turn probabilities of step t+1 fused into the link update of step t or
    launched on their own (PEDN_FUSE_TP), node_kernel unrolled for 6 or 8 corridors per node (PEDN_NODE_MD: 8 waves per SIMD / 6),
    the workgroup order inside link_turn_kernel, the owner-wave and single-launch plans.  Every plan, step-by-step stepping with a
    setter in between (which drops the fused probabilities) and a reset must give the same bits in every field; the stand-alone
    launch is the one the goldens above pin for large models."""
    from pednstream_amd import NetworkEnvGenerator
    from golden_util import DATA

    def history(fuse, stepwise, general="0", lds_limit="64", md="6", heavy="2", pairs_adj="1", owner="0", inline="0"):
        monkeypatch.setenv("PEDN_INLINE_TF", inline)          # 1: the single-launch plan -- node_kernel<LU, TF>'s slot waves compute their own rows (where the model's rows allow); 2: helper waves do
        monkeypatch.setenv("PEDN_LINK_OWNER", owner)          # 1: pedn_run's owner-wave plan -- node_kernel<LU>(t + 1) performs the link update of t
        monkeypatch.setenv("PEDN_PAIRS_ADJ", pairs_adj)        # 0: the link update takes its two link ids from the corridor's record (models whose
                                                               # directions are not numbered 2p, 2p + 1), 1: from the corridor index where the model allows
        monkeypatch.setenv("PEDN_TF_HEAVY_GROUPS", heavy)      # which rows of dynamic nodes go in front of the link update inside link_turn_kernel
        monkeypatch.setenv("PEDN_FUSE_TP", fuse)
        monkeypatch.setenv("PEDN_NODE_MD", md)                # 8: node_kernel unrolled for 8 corridors per node whatever the model has
        monkeypatch.setenv("PEDN_TF_GENERAL", general)        # 3: softmax groups and row sums through their general (any-size) paths
        monkeypatch.setenv("PEDN_TF_LDS_LIMIT", lds_limit)    # 1: all but one probability of a row overflow from LDS into HBM
        np.random.seed(7)
        net = NetworkEnvGenerator(DATA).create_network(name, verbose=False, n_replicas=64, rng_seed=11)
        if stepwise:
            link = next(iter(net.links.values()))
            for t in range(1, steps):
                net.network_loading(t)
                if t % 7 == 0:
                    link.back_gate_width = link.back_gate_width     # same value: only invalidates the fused probabilities
            net.engine().reset()                                      # second episode on the same engine
            for t in range(1, steps):
                net.network_loading(t)
        elif owner == "1":                                            # ranges of one, two and many steps, a single step in between
            net.run(1, 2)
            net.run(2, 4)
            net.network_loading(4)
            net.run(5, steps - 3)
            net.run(steps - 3, steps)
        else:
            net.run(1, steps)
        e = net.engine()
        out = {f: e.read_block(LINK_FIELDS[f][0], 0, steps) for f in ALL_FIELDS}
        net.close()
        return out

    ref = history("0", False)
    for variant in (("1", False), ("1", True), ("1", False, "3", "64"), ("0", False, "1", "1"),
                    ("1", True, "0", "0"), ("1", False, "0", "64", "8"), ("0", True, "0", "64", "8"),
                    ("1", False, "0", "64", "6", "0"), ("1", True, "0", "64", "6", "100"),
                    ("1", False, "0", "64", "6", "2", "0"), ("0", True, "0", "64", "6", "2", "0"),
                    ("1", False, "0", "64", "6", "2", "1", "1"), ("0", False, "0", "64", "8", "2", "0", "1"),
                    # step by step under the owner-wave plan: every network_loading(t) leaves its link update pending, the setter in
                    # between and the reads at the end perform it
                    ("1", True, "0", "64", "6", "2", "1", "1"), ("0", True, "0", "64", "8", "2", "1", "1"),
                    # the single-launch plan: ranges, single steps, step by step with setters and a reset
                    ("1", False, "0", "64", "6", "2", "1", "1", "1"), ("1", True, "0", "64", "6", "2", "1", "1", "1"),
                    ("0", False, "0", "64", "8", "2", "0", "1", "1"),
                    # ... with helper waves (node_kernel_h: sixteen waves per workgroup, the second eight compute the rows)
                    ("1", False, "0", "64", "6", "2", "1", "1", "2"), ("1", True, "0", "64", "6", "2", "1", "1", "2"),
                    ("0", False, "3", "1", "8", "2", "0", "1", "2")):
        got = history(*variant)
        for f in ALL_FIELDS:
            assert np.array_equal(ref[f], got[f]), (variant, f)


def test_repeated_step_after_imposed_fractions_recomputes_them_under_every_plan(monkeypatch):
    """update_turning_fractions_per_node on a dynamic node, then the LAST step again: the node recomputes its fractions at every
    network_loading (network.py:272-275), so the repeat must not take the imposed values that a read of the last step's buffer returns.
    Under the single-launch plan that buffer is the one the repeated step would reuse (found by tools/gpu_fuzz_plans.py, seed 950153)."""
    from pednstream_amd import NetworkEnvGenerator
    from golden_util import DATA

    def history(plain):
        for k in ("PEDN_LINK_OWNER", "PEDN_INLINE_TF"):
            monkeypatch.setenv(k, "0") if plain else monkeypatch.delenv(k, raising=False)
        np.random.seed(7)
        net = NetworkEnvGenerator(DATA).create_network("nine_intersections", verbose=False, n_replicas=64, rng_seed=11)
        nd = next(n for n in net.nodes.values() if len(n.incoming_links) > 2 and len(n.incoming_links) == len(n.outgoing_links))
        m = len(nd.incoming_links)
        for t in range(1, 30):
            net.network_loading(t)
        tf = np.random.default_rng(3).dirichlet(np.ones(m - 1), size=m).reshape(-1)
        net.update_turning_fractions_per_node([nd.node_id], [tf])
        seen = net.engine().get_turning_fractions(nd.index, 0).copy()      # the imposed values until the next step
        net.engine().step(29)                                               # the last step again
        for t in range(30, 60):
            net.network_loading(t)
        e = net.engine()
        out = {f: e.read_block(LINK_FIELDS[f][0], 0, 60) for f in ALL_FIELDS}
        info = e.plan_info()
        net.close()
        return out, seen, info

    ref, seen_ref, _ = history(True)
    got, seen, info = history(False)
    assert info["link_update_by_next_node_kernel"], info
    assert np.array_equal(seen, seen_ref)
    for f in ALL_FIELDS:
        assert np.array_equal(ref[f], got[f]), f


@pytest.mark.parametrize("name,steps1,steps2,R,streams", [("nine_intersections", 260, 90, 64, "1"), ("long_corridor", 200, 70, 64, "1"),
                                                          ("melbourne", 120, 40, 64, "1"), ("delft", 60, 25, 64, "1"),
                                                          ("melbourne", 90, 40, 256, "2"), ("nine_intersections", 150, 60, 256, "2")])
def test_lazy_reset_serves_a_second_episode_like_a_fresh_engine(name, steps1, steps2, R, streams, monkeypatch):
    """pedn_reset_lazy restores only what a new episode reads before it writes and declares every other row unwritten.  A second,
    DIFFERENT and shorter episode on the same engine must equal that episode on a fresh engine in every row of every field -- the rows
    behind its last step included, which still hold the first episode's values in memory: reads answer them with the initial values,
    get_outflow's wrapped look-backs likewise (nine_intersections: jammed links look back further than the episode is old).  Then a
    step that skips ahead, and a zero-copy consumer (device_ptr), after which the rows are really clear."""
    from pednstream_amd import NetworkEnvGenerator
    from golden_util import DATA

    monkeypatch.setenv("PEDN_STREAMS", streams)          # 2: the halves of the batch as two chains of launches (the catch-up clears
    monkeypatch.setenv("PEDN_STREAM_PROBE", "0")         # run before the fork, the bookkeeping once per step)

    def build():
        np.random.seed(7)
        return NetworkEnvGenerator(DATA).create_network(name, verbose=False, n_replicas=R, rng_seed=11)

    def demand(net, key, scale):
        T = net.simulation_steps
        for nid in net.origin_nodes:
            rows = np.stack([np.random.default_rng(key + 7 * r + 1000 * int(nid)).poisson(scale, T).astype(np.float64) for r in range(R)])
            net.set_demand_matrix(nid, rows)

    def everything(net, upto):
        e = net.engine()
        out = {f: e.read_block(LINK_FIELDS[f][0], 0, upto) for f in ALL_FIELDS}
        out["flags"] = e.error_flags()[1]
        return out

    a = build()
    T = a.simulation_steps
    demand(a, 1, 30.0)
    a.run(1, steps1, check=False)                                   # episode 1: busy, long
    a.reset(lazy=True)
    demand(a, 2, 6.0)
    a.run(1, steps2, check=False)                                   # episode 2: another demand, shorter
    b = build()
    demand(b, 2, 6.0)
    b.run(1, steps2, check=False)
    got, want = everything(a, T + 1), everything(b, T + 1)
    for f in want:
        assert np.array_equal(got[f], want[f]), f
    # skipping ahead after a lazy reset: the rows in between are cleared first
    a.reset(lazy=True)
    b.reset()
    # (the jumps stay inside the moving-average window, where avg_travel_time is travel_time[0] whether or not the step before ran: a
    # jump beyond it looks back zero steps -- PEDN_F_SAME_STEP, the reference raises there -- and reads the row being written)
    for net in (a, b):
        net.engine().step(3)
        net.engine().step(4)
        net.run(6, 22, check=False)                         # another jump, into a range long enough for two chains
    got, want = everything(a, T + 1), everything(b, T + 1)
    # (a skipped step leaves sending_flow = -1 behind: negative-flow flags, on both sides; where the thinner moving average of a short
    # link rounds to a zero look-back -- delft -- the step reads the row being written and only the flags are comparable)
    assert np.array_equal(got["flags"], want["flags"])
    if not (want["flags"] & 16).any():
        for f in want:
            assert np.array_equal(got[f], want[f]), f
    a.reset(lazy=True)                                      # and a jump beyond the window: the same flags on both sides
    b.reset()
    for net in (a, b):
        net.engine().step(steps2 + 3)
    assert np.array_equal(a.engine().error_flags()[1], b.engine().error_flags()[1]) and a.engine().error_flags()[1].any()
    # a zero-copy consumer: the engine finishes the clear, the rows are physically what a fresh engine holds
    a.reset(lazy=True)
    a.run(1, 12, check=False)
    e = a.engine()
    assert e.device_ptr(LINK_FIELDS["inflow"][0])
    b.reset()
    b.run(1, 12, check=False)
    got, want = everything(a, T + 1), everything(b, T + 1)
    for f in want:
        assert np.array_equal(got[f], want[f]), f
    a.close()
    b.close()


def test_full_size_melbourne_1024_invariants():
    """BASELINE config at full size (melbourne x 1024): size-independent properties of the model
    (SURVEY section 4): cumulative = running sum of flows, pedestrian conservation, non-negativity,
    outflow <= sending flow, inflow <= receiving flow, node flow conservation, replica independence."""
    from pednstream_amd import NetworkEnvGenerator
    from golden_util import DATA

    R, steps = 1024, 140
    net = NetworkEnvGenerator(DATA).create_network("melbourne", verbose=False, n_replicas=R, rng_seed=1)
    e = net.engine()
    # replicas 1000.. share demand with replica 0 but not its RNG key; make two replicas exact twins via the key offset
    net.run(1, steps)
    rc, _ = e.error_flags()
    assert rc == 0
    L = e.n_links
    rs = slice(0, 64)
    inflow = e.read_block(0, 0, steps, rep0=0, rep1=64)
    outflow = e.read_block(1, 0, steps, rep0=0, rep1=64)
    ci = e.read_block(2, 0, steps, rep0=0, rep1=64)
    co = e.read_block(3, 0, steps, rep0=0, rep1=64)
    S = e.read_block(4, 0, steps, rep0=0, rep1=64)
    Rv = e.read_block(5, 0, steps, rep0=0, rep1=64)
    N = e.read_block(9, 0, steps, rep0=0, rep1=64)
    assert np.array_equal(ci, np.cumsum(inflow, axis=0)) and np.array_equal(co, np.cumsum(outflow, axis=0))
    assert (inflow >= 0).all() and (outflow >= 0).all()
    assert np.array_equal(N.astype(np.float64), (ci - co)[:, :L])          # integer-valued counts: exact in f32
    assert (outflow[1:, :L] <= S[:-1] + 1e-9).all()
    assert (inflow[1:, :L] <= Rv[:-1] + 1e-9).all()
    m = flatten_network(net)
    for n in range(m["n_nodes"]):
        a, b = m["node_slot_ptr"][n], m["node_slot_ptr"][n + 1]
        q_out = outflow[:, m["slot_in_link"][a:b]].sum(axis=1)
        q_in = inflow[:, m["slot_out_link"][a:b]].sum(axis=1)
        assert np.array_equal(q_out, q_in), f"node {n}"
    assert ci[-1].sum() > 0                                                   # pedestrians actually entered
    # replicas differ (different RNG keys) ...
    d = e.read_block(10, steps - 1, steps)[0]
    assert not np.array_equal(d[:, 0], d[:, 1])
    # ... and one replica re-run alone with the same key reproduces itself exactly
    solo = NetworkEnvGenerator(DATA).create_network("melbourne", verbose=False, n_replicas=1, rng_seed=1, replica_offset=777)
    solo.run(1, steps)
    d_solo = solo._engine.read_block(10, steps - 1, steps)[0][:, 0]
    assert np.array_equal(d_solo, d[:, 777])
    solo.close()
    net.close()


def test_c_abi_argument_errors_and_reset():
    g = Golden("six_node_full")
    net = build_network(g, n_replicas=3)
    e = net.engine()
    lib = eng.lib()
    assert lib.pedn_step(e._h, 0) < 0 and b"outside" in lib.pedn_last_error(e._h)
    assert lib.pedn_set_width(e._h, 1, 10 ** 6, -1, 1.0) < 0
    assert lib.pedn_set_demand(e._h, net.nodes[0].index, -1, np.zeros(4).ctypes.data_as(C.POINTER(C.c_double)), 4) < 0  # node 0 has no virtual link
    with pytest.raises(IndexError):
        net.network_loading(10 ** 6)
    # demand entry points: bad node / replica / length / pattern are refused with a message, nothing is launched
    origin = next(n for n in net.nodes.values() if n.virtual_incoming_link is not None and n.node_id in net.origin_nodes)
    F, I = C.POINTER(C.c_double), C.POINTER(C.c_int32)
    buf = np.zeros(3 * (e.T + 2))
    assert lib.pedn_set_demand_matrix(e._h, net.nodes[0].index, buf.ctypes.data_as(F), 4) < 0
    assert lib.pedn_set_demand_matrix(e._h, origin.index, buf.ctypes.data_as(F), e.T + 2) < 0 and b"time indices" in lib.pedn_last_error(e._h)
    assert lib.pedn_get_demand(e._h, origin.index, 3, buf.ctypes.data_as(F), 4) < 0 and b"replica" in lib.pedn_last_error(e._h)
    pat, zi, zf = np.array([0, 7, 0], np.int32), np.zeros(3, np.int32), np.zeros(3)
    assert lib.pedn_draw_demand(e._h, origin.index, 1, pat.ctypes.data_as(I), zf.ctypes.data_as(F), zf.ctypes.data_as(F), zi.ctypes.data_as(I),
                                zi.ctypes.data_as(I), zf.ctypes.data_as(F)) < 0 and b"pattern" in lib.pedn_last_error(e._h)
    before = e.get_demand(origin.index, 1)
    assert np.array_equal(before[:len(origin.demand)], np.asarray(origin.demand, dtype=float))     # nothing was overwritten
    # pedn_set_demand_rows: a subset of replicas in one upload; the others keep what they had; the Network wrapper does not
    # re-broadcast the base demand afterwards (the node is no longer marked dirty)
    rows = np.arange(2 * 7, dtype=float).reshape(2, 7) + 1
    e.set_demand_rows(origin.index, [2, 0], rows)
    assert np.array_equal(e.get_demand(origin.index, 2)[:9], np.r_[rows[0], 0, 0]) and np.array_equal(e.get_demand(origin.index, 0)[:9], np.r_[rows[1], 0, 0])
    assert np.array_equal(e.get_demand(origin.index, 1), before)
    rep = np.array([5], np.int32)
    assert lib.pedn_set_demand_rows(e._h, origin.index, rep.ctypes.data_as(I), 1, buf.ctypes.data_as(F), 4) < 0 and b"replica" in lib.pedn_last_error(e._h)
    full = np.stack([before[:e.T + 1]] * 3)
    net.set_demand_matrix(origin.node_id, full)
    net.synchronize()
    assert np.array_equal(e.get_demand(origin.index, 2), before)
    net.run(1, 50)
    a = e.read_block(2, 0, 50)
    e.reset()
    assert (e.read_block(2, 0, 50) == 0).all() and (e.read_block(4, 0, 50) == -1).all()
    net.run(1, 50)
    assert np.array_equal(a, e.read_block(2, 0, 50))
    # negative demand trips the reference's "negative flows" check (node.py:218-219) -> sticky flag -> exception
    net.nodes[1].demand[60] = -5.0
    with pytest.raises(eng.ModelError):
        net.run(50, 70)
    net.close()


def test_link_view_methods_on_device_histories():
    """LinkView.get_density / get_outflow (link.py:190-214) evaluated on the host from fetched histories agree with the
    oracle's arithmetic (float32 scalar powers = glibc powf)."""
    g = Golden("nine_full")
    net = build_network(g)
    net.run(1, 120)
    L = od.lib()
    checked = 0
    for link in list(net.links.values())[:8]:
        inflow, att = np.asarray(link.inflow), np.asarray(link.avg_travel_time)
        N, Nr = np.asarray(link.num_pedestrians), np.asarray(link.reverse_link.num_pedestrians)
        for t in (40, 80, 119):
            tau = int(round(float(att[t]) / net.unit_time))
            F = np.float32(1.0) / (np.float32(1.0) + np.float32(link.gamma) * att[t])
            G = np.float32(1.0) - F
            G2, G3 = np.float32(L.pedn_oracle_powf(float(G), 2.0)), np.float32(L.pedn_oracle_powf(float(G), 3.0))
            want = (np.float64(F) * inflow[t - tau] + np.float64(F * G) * inflow[t - tau - 1] + np.float64(F * G2) * inflow[t - tau - 2]
                    + np.float64(F * G3) * inflow[t - tau - 3])
            assert link.get_outflow(t, tau) == max(np.ceil(want), 0)
            assert link.get_density(t) == (N[t] + Nr[t]) / np.float32(link.length * link.width)
            checked += 1
    assert checked == 24
    net.close()


def test_per_replica_widths_and_turning_fractions_via_replica_scope():
    """`with network.replica(r)` scopes setters to one replica (the replica-uniform fast path must fall back to the
    per-replica rows); every replica is checked against a CPU oracle run with that replica's settings."""
    g = Golden("forky")
    R, steps = 6, 120
    net = build_network(g, n_replicas=R, rng_seed=3)
    custom_tf = {2: np.array([0.2, 0.8, 0.5, 0.5, 0.9, 0.1]), 4: np.array([1.0, 0.0, 0.3, 0.7, 0.0, 1.0])}
    gate = {1: 0.4, 4: 0.0}
    model = flatten_network(net)
    base_tf = np.array(g.info["tf_values"][0])
    for r, tf in custom_tf.items():
        with net.replica(r):
            net.nodes[1].turning_fractions = tf
    net.run(1, 40)
    for r, w in gate.items():
        with net.replica(r):
            net.links[(1, 2)].back_gate_width = w
            assert net.links[(2, 1)].front_gate_width == w
    assert net.links[(1, 2)].back_gate_width == 1                      # replica 0 untouched
    net.run(40, steps)
    e = net._engine
    for r in range(R):
        o = od.Oracle(model, seed=3, replica=r)
        o.set_tf(net.nodes[1].index, custom_tf.get(r, base_tf))
        o.run(1, 40)
        if r in gate:
            o.set_width(1, net.links[(1, 2)].index, gate[r])
            o.set_width(0, net.links[(2, 1)].index, gate[r])
        o.run(40, steps)
        for fname in ALL_FIELDS:
            mine = e.read_block(LINK_FIELDS[fname][0], 0, steps, rep0=r, rep1=r + 1)[:, :, 0].T
            assert np.array_equal(mine[:e.n_links], o.field(fname)[:e.n_links, :steps]), (r, fname)
        with net.replica(r):
            assert np.array_equal(net.nodes[1].turning_fractions, custom_tf.get(r, base_tf))
    net.close()


@pytest.mark.parametrize("name,reps,hist", [("nine_intersections", 256, "full"), ("melbourne", 384, "full"), ("delft", 256, "recent")])
def test_step_by_step_loop_keeps_the_chains_forked_across_calls(name, reps, hist, monkeypatch):
    """The reference's calling sequence -- for t in range(1, T): network_loading(t) -- on a batch that steps as two chains: from the third
    consecutive call the halves of the replicas step on two streams and stay forked ACROSS the calls (pedn_step); a read, a setter or a
    reset in between joins them.  Same bits as one chain, call for call."""
    from pednstream_amd import NetworkEnvGenerator
    from golden_util import DATA

    def history(streams):
        monkeypatch.setenv("PEDN_STREAMS", streams)
        monkeypatch.setenv("PEDN_STREAM_PROBE", "0")
        np.random.seed(7)
        net = NetworkEnvGenerator(DATA).create_network(name, verbose=False, n_replicas=reps, rng_seed=5, history=hist)
        e = net.engine()
        assert e.plan_info()["chains"] == int(streams)
        out = []
        steps = 45
        for t in range(1, steps):
            net.network_loading(t)
            if t == 17:                                   # a read in the middle of the loop (joins the chains), then the loop goes on
                out.append(e.read_block(LINK_FIELDS["cumulative_inflow"][0], t, t + 1))
            if t == 29:                                   # a setter
                link = next(iter(net.links.values()))
                link.back_gate_width = 0.5 * link.width
        first = 0 if hist == "full" else steps - 3
        for f in ALL_FIELDS:
            last = steps - 2 if hist == "recent" and f in ("sending_flow", "receiving_flow") else steps - 1
            out.append(e.read_block(LINK_FIELDS[f][0], min(first, last - 1), last + 1))
        out.append(np.stack([np.concatenate([e.get_turning_fractions(nd.index, r) for nd in net.nodes.values()]) for r in (0, 127, 128, reps - 1)]))
        out.append(e.error_flags()[1])
        e.reset()                                         # ... and a second episode, stepped the same way
        for t in range(1, 12):
            net.network_loading(t)
        out.append(e.read_block(LINK_FIELDS["num_pedestrians"][0], 9, 12))
        net.close()
        return out

    one, two = history("1"), history("2")
    assert len(one) == len(two)
    for k, (a, b) in enumerate(zip(one, two)):
        assert np.array_equal(a, b), k


@pytest.mark.parametrize("name,steps,hist,reps", [("nine_intersections", 150, "full", 256), ("delft", 40, "full", 256), ("long_corridor", 150, "full", 256),
                                                  ("melbourne", 60, "full", 256), ("nine_intersections", 60, "recent", 256), ("delft", 40, "recent", 256),
                                                  # 128-replica segments that do not halve: the chains take 256 + 128 replicas
                                                  ("delft", 40, "full", 384), ("melbourne", 60, "full", 320), ("nine_intersections", 60, "recent", 384)])
@pytest.mark.parametrize("owner", ["0", "1"])
def test_two_stream_plan_gives_identical_histories(name, steps, hist, reps, owner, monkeypatch):
    """pedn_run launches the two halves of a large batch as two chains on two streams (replicas are independent; the default
    from 768 replicas).  Same bits as the one-stream plan in every field and every replica, also when the run is cut into
    several calls, continues after a setter, and after a reset; turning fractions and error flags included."""
    from pednstream_amd import NetworkEnvGenerator
    from golden_util import DATA

    def history(streams):
        monkeypatch.setenv("PEDN_STREAMS", streams)
        monkeypatch.setenv("PEDN_LINK_OWNER", owner if streams == "2" else "0")   # the reference side: one chain, two launches per step
        np.random.seed(7)
        net = NetworkEnvGenerator(DATA).create_network(name, verbose=False, n_replicas=reps, rng_seed=11, history=hist)
        e = net.engine()
        cut = steps // 3
        net.run(1, cut)                                   # two calls: the second one starts from fused turning fractions
        link = next(iter(net.links.values()))
        link.back_gate_width = link.back_gate_width       # a setter between the calls (drops the fused fractions)
        net.run(cut, steps)
        first = 0 if hist == "full" else steps - 2       # recent mode: most fields are short rings, compare what they still hold

        def held(f):     # sending / receiving flow of step t are entries t - 1: in a ring the newest entry is one behind
            last = steps - 1 if hist == "recent" and f in ("sending_flow", "receiving_flow") else steps
            return e.read_block(LINK_FIELDS[f][0], min(first, last - 2), last)

        out = {f: held(f) for f in ALL_FIELDS}
        out["tf"] = np.stack([np.concatenate([e.get_turning_fractions(nd.index, r) for nd in net.nodes.values()]) for r in (0, 127, 128, reps - 1)])
        out["flags"] = e.error_flags()[1]
        assert e.plan_info()["chains"] == int(streams)
        e.reset()
        net.run(1, steps)
        out2 = {f: held(f) for f in ALL_FIELDS}
        net.close()
        return out, out2

    a, a2 = history("1")
    b, b2 = history("2")
    for f in a:
        assert np.array_equal(a[f], b[f]), f
    for f in a2:
        assert np.array_equal(a2[f], b2[f]), f
        assert np.array_equal(a[f], a2[f]), f             # the second episode repeats the first


def test_profile_run_uses_and_reports_the_launch_plan(monkeypatch):
    """pedn_profile_run steps the simulation under pedn_run's plan with every launch timed: the plan it reports follows
    pedn_set_streams and the length of the range, and the histories equal an untimed run's.  (The two-launch plan: this small batch
    would otherwise step under the single-launch plan, one launch per step -- checked at the end.)"""
    from pednstream_amd import NetworkEnvGenerator
    from golden_util import DATA

    monkeypatch.setenv("PEDN_INLINE_TF", "0")
    np.random.seed(7)
    net = NetworkEnvGenerator(DATA).create_network("nine_intersections", verbose=False, n_replicas=256, rng_seed=11)
    e = net.engine()
    e.set_streams(2)
    ms, chains = e.profile_run(1, 40)
    assert chains == 2 and ms[1] > 0 and ms[2] > 0
    ms, chains = e.profile_run(40, 44)            # too short to pay for the fork and the join
    assert chains == 1
    rows, chains = e.profile_timeline(44, 56)     # the same, launch by launch: (step, chain, kind, start ms, end ms)
    assert chains == 2 and rows.shape == (12 * 2 * 2, 5) and set(rows[:, 1]) == {0.0, 1.0} and set(rows[:, 2]) == {1.0, 2.0}
    assert (rows[:, 4] > rows[:, 3]).all() and rows[0, 3] == 0.0
    for c in (0, 1):                              # within a chain the launches follow one another
        mine = rows[rows[:, 1] == c]
        assert (mine[1:, 3] >= mine[:-1, 4] - 1e-6).all()
    e.set_streams(1)
    ms, chains = e.profile_run(56, 80)
    assert chains == 1 and ms[1] > 0
    timed = {f: e.read_block(LINK_FIELDS[f][0], 0, 80) for f in ALL_FIELDS}
    e.reset()
    net.run(1, 80)
    for f in ALL_FIELDS:
        assert np.array_equal(timed[f], e.read_block(LINK_FIELDS[f][0], 0, 80)), f
    with pytest.raises(Exception):
        e.set_streams(3)
    net.close()
    # the single-launch plan: node_kernel<LU, TF> alone per step, one link update behind the range
    monkeypatch.setenv("PEDN_INLINE_TF", "1")
    np.random.seed(7)
    net = NetworkEnvGenerator(DATA).create_network("nine_intersections", verbose=False, n_replicas=256, rng_seed=11)
    e = net.engine()
    assert e.plan_info()["link_update_by_next_node_kernel"]
    net.run(1, 10, check=False)                   # (a check of the error flags would perform the pending link update)
    rows, chains = e.profile_timeline(10, 30)
    assert chains == 1 and rows.shape == (20 + 1, 5) and list(rows[:-1, 2]) == [1.0] * 20 and rows[-1, 2] == 2.0
    one = {f: e.read_block(LINK_FIELDS[f][0], 0, 30) for f in ALL_FIELDS}
    for f in ALL_FIELDS:                          # (sending / receiving flow of step t are entries t - 1: entry 29 is step 30's)
        assert np.array_equal(one[f][:29], timed[f][:29]), f
    net.close()


@pytest.mark.parametrize("fuse_tp,general,lds_limit", [("1", "0", "64"), ("0", "0", "64"), ("1", "3", "1")])
def test_fuzz_random_networks_engine_equals_oracle(fuse_tp, general, lds_limit, monkeypatch):
    """(with the next step's turn probabilities fused into the link update launch, and launched on their own: nine of the
    networks put separator links into softmax groups, whose density the fused launch re-derives)
    40 random scenarios (random trees + chords, all three fundamental diagrams, separators, controllers, activity,
    noise, odd time steps; the generator of the offline reference campaign oracle/fuzz_vs_reference.py): HIP engine and
    CPU oracle agree bit for bit on every field, every replica, including the sticky error flags."""
    import copy

    from fuzz_cases import random_case
    from pednstream_amd import Network

    monkeypatch.setenv("PEDN_FUSE_TP", fuse_tp)
    monkeypatch.setenv("PEDN_TF_GENERAL", general)
    monkeypatch.setenv("PEDN_TF_LDS_LIMIT", lds_limit)
    ran = 0
    for seed in range(3000, 3040):
        adj, params, origins, dests = random_case(seed)
        np.random.seed(seed)
        try:
            net = Network(adj, copy.deepcopy(params), origin_nodes=origins, destination_nodes=dests, verbose=False, n_replicas=3,
                          rng_seed=seed, replica_offset=seed % 5)
        except KeyError:
            continue                     # controller node on no OD path: the reference raises KeyError too
        model = flatten_network(net)
        T = params["simulation_steps"]
        net.run(1, T, check=False)
        e = net._engine
        _, flags = e.error_flags()
        for r in range(3):
            o = od.Oracle(model, seed=seed, replica=seed % 5 + r)
            o.run(1, T)
            assert int(flags[r]) == o.flags(), (seed, r, int(flags[r]), o.flags())
            if o.flags():
                continue                 # after a raise site the reference stops; values past it are not defined
            for fname in ALL_FIELDS:
                mine = e.read_block(LINK_FIELDS[fname][0], 0, T, rep0=r, rep1=r + 1)[:, :, 0].T
                assert np.array_equal(mine[:e.n_links], o.field(fname)[:e.n_links, :T]), (seed, r, fname)
            tf = np.concatenate([e.get_turning_fractions(nd.index, r) for nd in net.nodes.values()])
            assert np.array_equal(tf, o.tf()), (seed, r)
            ran += 1
        net.close()
    assert ran >= 60


@pytest.mark.parametrize("case", ["nine_full", "long_corridor_full", "delft_prefix"])
def test_recent_history_mode_gives_the_same_numbers(case):
    """PEDN_HIST_RECENT keeps inflow / cumulative_inflow whole and everything else as short rings: the complete arrays must
    equal the golden, and at several stops the entries still inside every ring must, too; older entries are refused."""
    g = Golden(case)
    net = build_network(g, n_replicas=2, replica_offset=g.replica, rng_seed=g.seed, rng_mode=g.mode, history="recent")
    e = net.engine()
    T1 = e.T + 1
    rows = {name: e.history_rows(LINK_FIELDS[name][0]) for name in ALL_FIELDS}
    assert rows["inflow"] == T1 and rows["cumulative_inflow"] == T1 and rows["sending_flow"] == 4 and rows["density"] == 4
    assert rows["travel_time"] < T1 and rows["cumulative_outflow"] < T1
    stops = sorted({min(g.steps - 1, s) for s in (3, 17, 60, g.steps - 1)})
    t = 1
    for stop in stops:
        while t <= stop:
            net.network_loading(t)
            for mut in g.mutations:
                if mut[0] == t:
                    apply_mutation(net, mut)
            t += 1
        for name in ALL_FIELDS:
            fid = LINK_FIELDS[name][0]
            newest = stop - 1 if name in ("sending_flow", "receiving_flow") else stop     # S / R of step t are entries t - 1
            lo = max(0, newest - rows[name] + 2) if rows[name] < T1 else 0
            mine = e.read_block(fid, lo, newest + 1)[:, :e.n_links, 0].T
            assert np.array_equal(mine, g.state(name)[:, lo:newest + 1]), (name, stop)
        lk = net.links[tuple(g.static("link_uv")[0])]
        assert lk.density[stop] == g.state("density")[0, stop]
        if stop > 10:
            with pytest.raises(IndexError):
                lk.density[stop - 6]
            col = np.asarray(lk.speed)
            assert np.isnan(col[stop - 6]) and col[stop] == g.state("speed")[0, stop]
            # sending_flow[stop] is written by step stop + 1: its ring slot still holds entry stop - 4, which must not be served
            with pytest.raises(IndexError):
                lk.sending_flow[stop]
            col = np.asarray(lk.sending_flow)
            assert np.isnan(col[stop]) and col[stop - 1] == g.state("sending_flow")[0, stop - 1]
    assert e.error_flags()[0] == 0
    net.close()


def test_od_weights_changed_between_steps_take_effect_at_the_next_step():
    """The turning fractions of step t + 1 are computed in the launch behind node_kernel(t); a setter in between that they
    depend on (OD weights, back gate widths) must make the engine recompute them -- compared with the oracle doing the same."""
    g = Golden("nine_full")
    net = build_network(g, n_replicas=2, replica_offset=g.replica, rng_seed=g.seed)
    model = flatten_network(net)
    e = net.engine()
    o = od.Oracle(model, seed=g.seed, replica=g.replica)
    T1 = e.T + 1
    rng = np.random.default_rng(5)
    P = C.POINTER(C.c_double)
    for t in range(1, 90):
        net.network_loading(t)
        o.step(t)
        if t in (20, 21, 55):
            for k in range(int(model["n_od"])):
                w = rng.uniform(0.0, 10.0, T1)
                e.set_od_weights(k, w)
                o.L.pedn_oracle_set_od_weights(o.h, k, np.ascontiguousarray(w).ctypes.data_as(P), T1)
    for name in ALL_FIELDS:
        mine = e.read_block(LINK_FIELDS[name][0], 0, 90)[:, :e.n_links, 0].T
        assert np.array_equal(mine, o.field(name)[:e.n_links, :90]), name
    tf = np.concatenate([e.get_turning_fractions(nd.index, 0) for nd in net.nodes.values()])
    assert np.array_equal(tf, o.tf())
    assert not np.array_equal(o.field("inflow")[:e.n_links, :90], g.state("inflow")[:, :90])      # the change mattered
    net.close()


@pytest.mark.parametrize("name", ["melbourne", "nine_intersections"])
def test_zero_copy_pointer_held_across_steps_and_lazy_resets_needs_pedn_flush(name):
    """a consumer that fetched pedn_device_ptr ONCE and orders itself behind pedn_stream() sees complete rows only after
    pedn_flush (or a fresh pedn_device_ptr / pedn_synchronize): under the owner-wave / single-launch plans the last step's link update is
    pending after pedn_run, and after pedn_reset_lazy the rows above the current step still hold the previous episode.  The stream-ordered
    read through the held pointer must equal pedn_read after pedn_flush -- and the pointer must still be the same."""
    torch = pytest.importorskip("torch")
    from pednstream_amd import NetworkEnvGenerator
    from golden_util import DATA
    from pednstream_amd.rl_env import _DeviceBuffer

    np.random.seed(7)
    R = 64
    net = NetworkEnvGenerator(DATA).create_network(name, verbose=False, n_replicas=R, rng_seed=11)
    e = net.engine()
    assert e.plan_info()["link_update_by_next_node_kernel"]
    T1 = e.T + 1
    views = {}
    for f in ("density", "num_pedestrians", "speed", "cumulative_inflow"):
        fid = LINK_FIELDS[f][0]
        ptr, cols, stride = e.device_ptr(fid)
        views[f] = (fid, ptr, torch.as_tensor(_DeviceBuffer(ptr, (T1, cols, stride), "<f8" if fid < 7 else "<f4"), device="cuda"))
    ext = torch.cuda.ExternalStream(e.stream_ptr())

    def through_pointer(f, upto):
        fid, ptr, view = views[f]
        assert e.device_ptr(fid)[0] == ptr
        torch.cuda.current_stream().wait_stream(ext)          # stream order only, no host synchronisation of the engine
        return view[:upto, :e.n_links, :R].cpu().numpy()

    net.run(1, 40, check=False)
    e.flush()
    for f in views:
        assert np.array_equal(through_pointer(f, 40), e.read_block(views[f][0], 0, 40)[:, :e.n_links]), f
    assert through_pointer("density", 40)[39].any() or name == "melbourne"          # the last step's link update is there
    net.reset(lazy=True)
    net.run(1, 12, check=False)
    e.flush()
    for f in views:
        got = through_pointer(f, T1)
        assert np.array_equal(got, e.read_block(views[f][0], 0, T1)[:, :e.n_links]), f
        assert not got[13:].any(), f                           # the previous episode's rows are physically gone
    net.close()


def test_bin_packing_by_degree_static_estimate_and_measured_cost_gives_identical_histories(monkeypatch):
    """Which nodes share a node-kernel workgroup (PEDN_PACK_BY_LOAD: 0 by degree, 1 by the static load estimate, 2 by the measured node
    cost of data/melbourne/pack_cost.json -- the default when the scenario carries one) only decides how long a workgroup waits at its
    barriers: every field, flag and fraction is the same."""
    from pednstream_amd import NetworkEnvGenerator
    from golden_util import DATA

    def history(pack):
        monkeypatch.delenv("PEDN_PACK_BY_LOAD", raising=False) if pack is None else monkeypatch.setenv("PEDN_PACK_BY_LOAD", pack)
        np.random.seed(7)
        net = NetworkEnvGenerator(DATA).create_network("melbourne", verbose=False, n_replicas=64, rng_seed=11)
        for nid in net.origin_nodes:
            net.set_demand_matrix(nid, np.stack([np.random.default_rng(5 + r).poisson(60.0, net.simulation_steps).astype(np.float64) for r in range(64)]))
        net.run(1, 80)
        e = net.engine()
        out = {f: e.read_block(LINK_FIELDS[f][0], 0, 80) for f in ALL_FIELDS}
        out["flags"] = e.error_flags()[1]
        packed = e.plan_info()["packed_by"]
        net.close()
        return out, packed

    ref, how = history(None)
    assert how == "measured_node_cost"
    for pack, name in (("0", "degree"), ("1", "static_load_estimate")):
        got, how = history(pack)
        assert how == name
        for f in ref:
            assert np.array_equal(ref[f], got[f]), (pack, f)
    assert ref["cumulative_inflow"][79].sum() > 0
