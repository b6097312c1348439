"""The BASELINE configurations at their FULL batch sizes, under the launch plan the engine picks by itself (no PEDN_STREAMS /
PEDN_FUSE_* override): delft x 1024 (config #3), 45_intersections x 2048 plain and as the batched RL step (config #5, shared
and per-replica randomised scenarios).  Each test proves which plan ran (pedn_profile_run reports the number of chains) and
compares replicas from both halves of the two-chain split -- first, last of the first half, first of the second half, last --
bit for bit with the CPU oracle, plus the size-independent invariants of test_full_size_melbourne_1024_invariants on a
64-replica slab.  Also: stepping a dynamic-fraction model through the LAST time index (t = T)."""
import os

import numpy as np
import pytest

import oracle_driver as od
from golden_util import ALL_FIELDS, DATA, Golden, build_network
from pednstream_amd import NetworkEnvGenerator
from pednstream_amd.flatten import flatten_network
from pednstream_amd.network import LINK_FIELDS
from rl_oracle import RlOracle
from scenario_models import replica_model

pytestmark = pytest.mark.gpu


def poisson_demand(T, key, base=5.0, peak=10.0):
    """bench.py's per-replica origin demand (od_manager.py:145-155 shape)"""
    t = np.arange(T)
    lam = base + peak * np.exp(-(t - T / 4) ** 2 / (2 * (T / 20) ** 2)) + peak * np.exp(-(t - 3 * T / 4) ** 2 / (2 * (T / 20) ** 2))
    return np.random.default_rng(1000 + key).poisson(lam).astype(np.float64)


def invariants(e, model, steps, r0, r1):
    """SURVEY section 4 on replicas [r0, r1): cumulative = running sum, conservation per link and per node, bounds."""
    L = e.n_links
    inflow, outflow = (e.read_block(f, 0, steps, rep0=r0, rep1=r1) for f in (0, 1))
    ci, co = (e.read_block(f, 0, steps, rep0=r0, rep1=r1) for f in (2, 3))
    S, Rv = (e.read_block(f, 0, steps, rep0=r0, rep1=r1) for f in (4, 5))
    N = e.read_block(9, 0, steps, rep0=r0, rep1=r1)
    # non-integer flows exist on delft (SURVEY headline fact 4): the running sums are compared as the device formed them
    assert np.array_equal(ci[1:], ci[:-1] + inflow[1:]) and np.array_equal(co[1:], co[:-1] + outflow[1:])
    assert (inflow >= 0).all() and (outflow >= 0).all()
    assert np.allclose(N.astype(np.float64), (ci - co)[:, :L], rtol=1e-6, atol=1e-4)
    assert (outflow[1:, :L] <= S[:-1] + 1e-9).all() and (inflow[1:, :L] <= Rv[:-1] + 1e-9).all()
    for n in range(model["n_nodes"]):
        a, b = model["node_slot_ptr"][n], model["node_slot_ptr"][n + 1]
        # classic rule: both sides are sums of the same floored matrix (node.py:298); one-to-one: the same two numbers
        assert np.array_equal(outflow[:, model["slot_in_link"][a:b]].sum(axis=1), inflow[:, model["slot_out_link"][a:b]].sum(axis=1)), n
    assert ci[-1].sum() > 0


@pytest.mark.parametrize("name,R,steps", [("delft", 1024, 130), ("45_intersections", 2048, 120)])
def test_full_size_batch_under_the_default_plan(name, R, steps):
    for k in ("PEDN_STREAMS", "PEDN_FUSE_TP", "PEDN_LINK_OWNER", "PEDN_INLINE_TF"):
        assert k not in os.environ, f"{k} is set: this test is about the plan the engine picks by itself"
    np.random.seed(7)
    net = NetworkEnvGenerator(DATA).create_network(name, verbose=False, n_replicas=R, rng_seed=3)
    e = net.engine()
    T = net.simulation_steps
    origins = list(net.origin_nodes)
    for nid in origins:
        net.set_demand_matrix(nid, np.stack([poisson_demand(T, 7 * r + nid) for r in range(R)]))
    cut = steps - 24
    net.run(1, cut)
    ms, chains = e.profile_run(cut, steps)          # the same plan pedn_run uses, with every launch timed
    assert chains == 2, "models with dynamic turning-fraction rows step as two chains of launches from 1024 replicas"
    assert ms[1] > 0 and ms[2] > 0
    rc, _ = e.error_flags()
    assert rc == 0
    model = flatten_network(net)
    invariants(e, model, steps, R // 2 - 32, R // 2 + 32)          # a slab across the split
    for r in (0, R // 2 - 1, R // 2, R - 1):
        o = od.Oracle(model, seed=3, replica=r)
        for nid in origins:
            o.set_demand(net.nodes[nid].index, poisson_demand(T, 7 * r + nid))
        o.run(1, steps)
        assert o.flags() == 0
        for fname in ALL_FIELDS:
            mine = e.read_block(LINK_FIELDS[fname][0], 0, steps, rep0=r, rep1=r + 1)[:, :, 0].T
            cols = e.n_all if LINK_FIELDS[fname][0] < 4 else e.n_links          # the virtual links' flow arrays too
            assert np.array_equal(mine[:cols], o.field(fname)[:cols, :steps]), (r, fname)
        tf = np.concatenate([e.get_turning_fractions(nd.index, r) for nd in net.nodes.values()])
        assert np.array_equal(tf, o.tf()), r
        o.close()
    net.close()


def test_largest_single_gpu_batch_melbourne_4608():
    """More than BASELINE config #4's whole ensemble on ONE GPU (melbourne x 4608 replicas, 173 GB of histories): the flow and
    cumulative-count fields hold more than 2^31 ELEMENTS, every field more than 2^32 bytes -- any 32-bit index would show.  First,
    middle and last replica bit-exact against the oracle, no error flag."""
    R, steps = 4608, 500                   # element index (t * columns + column) * replicas + replica passes 2^31 at t = 476
    np.random.seed(7)
    net = NetworkEnvGenerator(DATA).create_network("melbourne", verbose=False, n_replicas=R, rng_seed=5)
    try:
        e = net.engine()
    except RuntimeError as err:          # a GPU with less than 173 GB free cannot hold this batch: that is not a parity failure
        if "not enough HBM" in str(err):
            pytest.skip(str(err))
        raise
    T = net.simulation_steps
    assert (steps - 1) * e.n_all * R > 2 ** 31 and steps == T
    origins = list(net.origin_nodes)
    for nid in origins:
        net.set_demand_matrix(nid, np.stack([poisson_demand(T, r, base=40.0, peak=80.0) for r in range(R)]))
    net.run(1, steps)
    rc, _ = e.error_flags()
    assert rc == 0
    model = flatten_network(net)
    for r in (0, R // 2, R - 1):
        o = od.Oracle(model, seed=5, replica=r)
        for nid in origins:
            o.set_demand(net.nodes[nid].index, poisson_demand(T, r, base=40.0, peak=80.0))
        o.run(1, steps)
        assert o.flags() == 0
        for fname in ALL_FIELDS:
            mine = e.read_block(LINK_FIELDS[fname][0], 0, steps, rep0=r, rep1=r + 1)[:, :, 0].T
            cols = e.n_all if LINK_FIELDS[fname][0] < 4 else e.n_links
            assert np.array_equal(mine[:cols], o.field(fname)[:cols, :steps]), (r, fname)
        o.close()
    assert e.read_block(2, steps - 1, steps, rep0=R - 1, rep1=R).sum() > 0
    net.close()


@pytest.mark.parametrize("randomized", [False, True])
def test_config5_rl_step_2048_envs_full_size(randomized):
    """45_intersections x 2048 envs, option3, 110 env steps of uniform random actions: observations and rewards of four envs
    against the restated RL glue on the CPU oracle.  randomized: after reset(options={'randomize': True}) every env carries
    its own link parameters, OD weights and demand (node_kernel<PR>, link_turn_kernel<PR, OBS>)."""
    from pednstream_amd.rl_env import VecPedNetEnv

    B, steps = 2048, 110
    np.random.seed(3)
    env = VecPedNetEnv("45_intersections", n_envs=B, obs_mode="option3", data_dir=DATA, seed=9)
    net = env.network
    e = net.engine()
    if randomized:
        obs0, _ = env.reset(options={"randomize": True}, seed=123)
        sc = env.scenarios
        assert len({tuple(sc.kc[:, r]) for r in (0, 1, B - 1)}) == 3
    else:
        obs0, _ = env.reset()
    model = flatten_network(net)
    spec = [{"id": aid, "type": env.agent_manager.get_agent_type(aid),
             "links": [l.link_id for l in (env.agent_manager.get_separator_links(aid) if env.agent_manager.get_agent_type(aid) == "sep"
                                           else env.agent_manager.get_gater_outgoing_links(aid))]} for aid in env.possible_agents]
    checks = {}
    for r in (0, B // 2 - 1, B // 2, B - 1):
        mr = replica_model(model, net, env.scenarios, r) if randomized else model
        kw = dict(link_kc=mr["link_kc"], link_kj=mr["link_kj"]) if randomized else {}
        checks[r] = RlOracle(net, mr, spec, "option3", False, 1, seed=9, replica=r, **kw)
    rng = np.random.default_rng(11)
    for k in range(steps):
        acts = rng.uniform(-0.5, 4.5, size=(B, env.n_actions)).astype(np.float32)     # ActionApplier sees float32 widths
        obs, rew, term, trunc, _ = env.step(acts)
        for r, orc in checks.items():
            o, w = orc.step(acts[r])
            assert np.array_equal(obs[r], o), (k, r)
            assert np.array_equal(rew[r], w), (k, r)
    e.check_errors()
    assert len({obs[r].tobytes() for r in checks}) > 1
    last = env.sim_step - 1
    for r, orc in checks.items():              # and the histories behind the observations
        for fname in ("cumulative_inflow", "cumulative_outflow", "density", "travel_time", "receiving_flow"):
            mine = e.read_block(LINK_FIELDS[fname][0], 0, last, rep0=r, rep1=r + 1)[:, :, 0].T
            assert np.array_equal(mine[:e.n_links], orc.o.field(fname)[:e.n_links, :last]), (r, fname)
    env.close()


def test_stepping_through_the_last_time_index():
    """pedn_step / pedn_run / pedn_rl_step accept t = T (the histories have T + 1 entries; VecPedNetEnv steps until
    sim_step == simulation_steps).  Behind that step no turning fractions of T + 1 are computed: the per-step tables end at T."""
    from pednstream_amd.rl_env import VecPedNetEnv

    g = Golden("nine_full")
    net = build_network(g, n_replicas=256, rng_seed=g.seed)       # 256: enough for the two-chain plan below
    e = net.engine()
    T = net.simulation_steps
    net.run(1, T - 3)
    for t in range(T - 3, T + 1):
        net.network_loading(t)
    model = flatten_network(net)
    for r in (0, 255):
        o = od.Oracle(model, seed=g.seed, replica=r)
        o.run(1, T + 1)
        for fname in ALL_FIELDS:
            mine = e.read_block(LINK_FIELDS[fname][0], 0, T + 1, rep0=r, rep1=r + 1)[:, :, 0].T
            assert np.array_equal(mine[:e.n_links], o.field(fname)[:e.n_links]), (r, fname)
    last_rows = {f: e.read_block(LINK_FIELDS[f][0], T - 2, T + 1) for f in ALL_FIELDS}
    for plan in (1, 2):                             # the same through pedn_run, as one chain and as two chains of launches
        e.set_streams(plan)
        e.reset()
        net.run(1, T - 10)
        assert e.profile_run(T - 10, T + 1)[1] == plan          # the range that ends with step T really ran under that plan
        assert e.error_flags()[0] == 0
        for f in ALL_FIELDS:
            assert np.array_equal(e.read_block(LINK_FIELDS[f][0], T - 2, T + 1), last_rows[f]), (plan, f)
    with pytest.raises(Exception):
        e.run(1, T + 2)
    net.close()
    env = VecPedNetEnv("nine_intersections", n_envs=64, obs_mode="option3", data_dir=DATA, seed=1)
    env.reset()
    term = False
    n = 0
    while not term:
        obs, rew, term, _, _ = env.step(None)
        n += 1
    assert n == env.simulation_steps and np.isfinite(obs).all()
    with pytest.raises(IndexError):
        env.step(None)
    env.close()
