"""Per-replica scenarios on one topology (rank 2): replicas of ONE engine carry different link parameters, OD weights
and demand; each replica must equal the reference run of its own scenario (goldens rand_*.npz, built by the reference's
create_network with the overrides its own randomisers produced)."""
import numpy as np
import pytest

import oracle_driver as od
from golden_util import ALL_FIELDS, DATA, Golden, compare_fields
from pednstream_amd import NetworkEnvGenerator
from pednstream_amd.flatten import flatten_network
from pednstream_amd.network import LINK_FIELDS
from pednstream_amd.scenarios import ScenarioBatch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,cases", [("nine_intersections", ["rand_nine_a", "rand_nine_b"]), ("delft", ["rand_delft_a", "rand_delft_b"])])
def test_replicas_with_different_scenarios_match_their_reference_runs(name, cases):
    goldens = [Golden(c) for c in cases]
    R = len(goldens) + 1                                   # last replica keeps the base scenario
    np.random.seed(goldens[0].info["np_seed"])
    gen = NetworkEnvGenerator(DATA)
    net = gen.create_network(name, verbose=False, n_replicas=R, rng_seed=0)
    base_demand = {nid: np.array(n.demand, dtype=float) for nid, n in net.nodes.items() if n.demand is not None}
    batch = ScenarioBatch(net, edge_distances=gen.network_data["edge_distances"])
    T = net.simulation_steps
    for r, g in enumerate(goldens):
        rz = g.info["randomized"]
        od_flows = {tuple(int(x) for x in k.split("_")): w for k, w in rz["od_flows"].items()}
        batch.set_replica(r, link_params_overrides=rz["link_params_overrides"], od_flows=od_flows, demand=g.demand())
    batch.commit()
    steps = min(g.steps for g in goldens)
    net.run(1, steps)
    e = net._engine
    rc, _ = e.error_flags()
    assert rc == 0
    for r, g in enumerate(goldens):
        problems = compare_fields(lambda nm: e.read_block(LINK_FIELDS[nm][0], 0, steps, rep0=r, rep1=r + 1)[:, :, 0].T, g, e.n_links, steps)
        assert not problems, f"replica {r} ({g.case}):\n" + "\n".join(problems)
        tf = np.concatenate([e.get_turning_fractions(nd.index, r) for nd in net.nodes.values()])
        assert np.array_equal(tf, g.z["tf_hist"][steps - 2]), f"turning fractions of replica {r}"
    # the replica without overrides equals the CPU oracle of the base scenario
    base = od.Oracle(flatten_network(net), seed=0, replica=R - 1)
    for nid, d in base_demand.items():
        base.set_demand(net.nodes[nid].index, d)
    base.run(1, steps)
    for nm in ALL_FIELDS:
        mine = e.read_block(LINK_FIELDS[nm][0], 0, steps, rep0=R - 1, rep1=R)[:, :, 0].T
        assert np.array_equal(mine[:e.n_links], base.field(nm)[:e.n_links, :steps]), nm
    net.close()


def test_scenario_batch_rejects_geometry_overrides_and_time_varying_weights():
    net = NetworkEnvGenerator(DATA).create_network("nine_intersections", verbose=False, n_replicas=2)
    batch = ScenarioBatch(net)
    with pytest.raises(ValueError):
        batch.set_replica(0, link_params_overrides={"0_1": {"length": 80}})
    with pytest.raises(ValueError):
        batch.set_replica(0, od_flows={(0, 8): np.arange(net.simulation_steps + 1.0)})
    net.close()


def test_vec_env_randomized_reset_gives_each_env_its_own_scenario():
    """reset(options={'randomize': True}): every env gets its own link parameters, OD weights and demand."""
    from pednstream_amd.rl_env import VecPedNetEnv

    B, steps = 8, 60
    np.random.seed(3)
    env = VecPedNetEnv("45_intersections", n_envs=B, obs_mode="option3", data_dir=DATA, seed=9)
    obs0, _ = env.reset(options={"randomize": True}, seed=123)
    sc = env.scenarios
    assert len({tuple(sc.kc[:, r]) for r in range(B)}) > 1 and len({tuple(sc.od_w[:, r]) for r in range(B)}) == B
    rng = np.random.default_rng(0)
    for _ in range(steps):
        obs, rew, term, trunc, _ = env.step(rng.uniform(0, 4, size=(B, env.n_actions)))
    assert np.isfinite(obs).all() and len({obs[r].tobytes() for r in range(B)}) > 1
    env.close()


def test_demand_drawn_on_device_has_the_reference_distributions():
    """pedn_draw_demand (od_manager.py:92-155 on the device, own Philox keys): constant rows exact, Poisson rows with mean
    and variance = lambda(t), the sudden-demand spike added exactly, reproducible per (seed, replica)."""
    R = 2048
    np.random.seed(1)
    net = NetworkEnvGenerator(DATA).create_network("nine_intersections", verbose=False, n_replicas=R, rng_seed=0)
    e = net.engine()
    T = net.simulation_steps
    node = next(n for n in net.nodes.values() if n.virtual_incoming_link is not None and n.node_id in net.origin_nodes)
    base, peak = np.full(R, 6.0), np.full(R, 20.0)
    zeros_i, zeros_f = np.zeros(R, dtype=np.int32), np.zeros(R)
    e.draw_demand(node.index, 77, np.zeros(R, np.int32), base, peak, zeros_i, zeros_i, zeros_f)
    gauss = np.stack([e.get_demand(node.index, r) for r in range(0, R, 2)])                     # [R/2, T+1]
    t = np.arange(T)
    lam = 6.0 + 20.0 * (np.exp(-(t - T / 4) ** 2 / (2 * (T / 20) ** 2)) + np.exp(-(t - 3 * T / 4) ** 2 / (2 * (T / 20) ** 2)))
    n = gauss.shape[0]
    assert np.all(gauss[:, T] == 0) and np.all(gauss == np.floor(gauss)) and gauss.min() >= 0
    z = (gauss[:, :T].mean(axis=0) - lam) / np.sqrt(lam / n)
    assert np.abs(z).max() < 5.0, np.abs(z).max()                                               # 500 time indices
    ratio = gauss[:, :T].var(axis=0, ddof=1) / lam                                              # Poisson: variance = mean
    assert 0.75 < ratio.min() and ratio.max() < 1.3 and abs(ratio.mean() - 1) < 0.02, (ratio.min(), ratio.max(), ratio.mean())
    assert len({gauss[i].tobytes() for i in range(n)}) == n                                      # replicas are independent draws
    # same seed, spike pattern: the same Poisson draws plus the spike
    start, length, height = np.full(R, 100, np.int32), np.full(R, 15, np.int32), np.full(R, 33.0)
    pattern = np.full(R, 2, np.int32)
    pattern[1::2] = 1                                                                           # odd replicas: constant
    e.draw_demand(node.index, 77, pattern, base, peak, start, length, height)
    spike = np.stack([e.get_demand(node.index, r) for r in range(0, R, 2)])
    diff = spike - gauss
    assert np.all(diff[:, 100:115] == 33.0) and np.all(np.delete(diff, np.s_[100:115], axis=1) == 0)
    assert np.array_equal(e.get_demand(node.index, 5), np.full(T + 1, 6.0))
    e.draw_demand(node.index, 78, np.zeros(R, np.int32), base, peak, zeros_i, zeros_i, zeros_f)
    assert not np.array_equal(e.get_demand(node.index, 0), gauss[0])
    # the per-replica matrix upload and the getter agree
    mat = np.random.default_rng(0).integers(0, 30, (R, T)).astype(np.float64)
    e.set_demand_matrix(node.index, mat)
    for r in (0, 1, R - 1):
        assert np.array_equal(e.get_demand(node.index, r), np.concatenate([mat[r], [0.0]]))
    net.close()


def test_vectorised_randomisation_draws_the_reference_ranges_for_every_env():
    """reset(options={'randomize': True, 'mode': 'vectorised'}): link parameters of exactly 20 % of the corridors change,
    inside the ranges of generate_random_link_params; OD weights in [1, 10]; demand drawn on the device; the run is finite."""
    from pednstream_amd.rl_env import VecPedNetEnv

    B, steps = 256, 40
    np.random.seed(3)
    env = VecPedNetEnv("45_intersections", n_envs=B, obs_mode="option3", data_dir=DATA, seed=9)
    env.reset(options={"randomize": True, "mode": "vectorised"}, seed=5)
    sc = env.scenarios
    kc0, kj0, vf0 = (sc._base[k][:, None] for k in ("kc", "kj", "vf"))
    assert sc.kc.shape == (env.network.n_links, B)
    assert np.all(sc.kc >= np.minimum(0.5, kc0)) and np.all(sc.kc <= kc0 * 1.2 + 1e-12)
    changed_kc = sc.kc != kc0
    assert np.all(sc.kj[changed_kc] >= 2.0 * sc.kc[changed_kc] - 1e-12) and np.all((sc.kj == kj0)[~changed_kc])
    assert np.all(sc.vf <= vf0) and np.all(sc.vf >= vf0 * 0.6 - 1e-12)
    pairs = len(sc._pair_links)
    k = int(pairs * 0.2)
    touched = (changed_kc | (sc.vf != vf0))
    per_env_pairs = np.array([len({frozenset((l.start_node.node_id, l.end_node.node_id)) for l in env.network._link_list if touched[l.index, r]})
                              for r in range(B)])
    assert per_env_pairs.max() <= k and 0.55 * k < per_env_pairs.mean() < 0.95 * k          # P(corridor changes | chosen) = 3/4
    assert sc.od_w.min() >= 1.0 and sc.od_w.max() <= 10.0 and len({tuple(sc.od_w[:, r]) for r in range(B)}) == B
    from pednstream_amd.scenarios import derive_statics
    l0 = env.network._link_list[0]
    for r in (0, B - 1):
        tt0, fft, tsw = derive_statics(l0.length, sc.vf[l0.index, r], sc.kc[l0.index, r], sc.kj[l0.index, r], env.network.unit_time)
        assert (tt0, fft, tsw) == (sc.tt0[l0.index, r], sc.fft[l0.index, r], sc.tau_sw[l0.index, r])
    e = env.network.engine()
    origin = next(n for n in env.network.nodes.values() if n.virtual_incoming_link is not None and n.node_id in env.network.origin_nodes)
    rows = np.stack([e.get_demand(origin.index, r) for r in range(B)])
    assert len({rows[r].tobytes() for r in range(B)}) == B and rows.min() >= 0 and rows[:, :-1].mean() > 2.0
    rng = np.random.default_rng(0)
    for _ in range(steps):
        obs, rew, term, trunc, _ = env.step(rng.uniform(0, 4, size=(B, env.n_actions)))
    e.check_errors()
    assert np.isfinite(obs).all() and np.isfinite(rew).all()
    env.close()


def test_two_chains_of_launches_with_per_replica_scenarios_and_with_the_node_lp(monkeypatch):
    """The two-stream plan of pedn_run (tests/test_gpu_parity.py) through the kernel instantiations it does not reach there:
    per-replica link parameters / OD weights / demand (node_kernel<PR>, link_kernel_1r<PR>, turn_frac_kernel<PR>) and the node LP
    (its tableau workspace is indexed by replica group).  Same bits as one chain of launches."""
    from pednstream_amd import NetworkEnvGenerator
    from pednstream_amd.network import LINK_FIELDS
    from pednstream_amd.rl_env import VecPedNetEnv

    def scenario_run(streams):
        monkeypatch.setenv("PEDN_STREAMS", streams)
        np.random.seed(3)
        env = VecPedNetEnv("45_intersections", n_envs=256, obs_mode="option3", data_dir=DATA, seed=9)
        env.reset(options={"randomize": True, "mode": "vectorised"}, seed=5)
        env.network.run(1, 60)
        e = env.network.engine()
        out = {f: e.read_block(LINK_FIELDS[f][0], 0, 60) for f in LINK_FIELDS}
        out["flags"] = e.error_flags()[1]
        env.close()
        return out

    def lp_run(streams):
        monkeypatch.setenv("PEDN_STREAMS", streams)
        gen = NetworkEnvGenerator(DATA)
        np.random.seed(7)
        gen.network_data = gen.load_network_data("nine_intersections")
        gen.config["params"]["assign_flows_type"] = "optimal"
        net = gen.create_network("nine_intersections", verbose=False, n_replicas=256, rng_seed=11)
        assert net.assign_flows_type == "optimal"
        net.run(1, 60, check=False)
        e = net.engine()
        out = {f: e.read_block(LINK_FIELDS[f][0], 0, 60) for f in LINK_FIELDS}
        out["flags"] = e.error_flags()[1]
        net.close()
        return out

    for run in (scenario_run, lp_run):
        a, b = run("1"), run("2")
        for f in a:
            assert np.array_equal(a[f], b[f]), (run.__name__, f)


@pytest.mark.parametrize("mode", ["reference", "vectorised"])
def test_randomised_reset_is_a_function_of_the_seed(mode):
    """Same seed -> the same scenarios, demand and trajectories (on a fresh env and on a reused one); another seed -> others."""
    from pednstream_amd.rl_env import VecPedNetEnv

    B, steps = 16, 30

    def episode(env, seed):
        obs0, _ = env.reset(options={"randomize": True, "mode": mode}, seed=seed)
        sc = env.scenarios
        e = env.network.engine()
        origin = next(n for n in env.network.nodes.values() if n.virtual_incoming_link is not None and n.node_id in env.network.origin_nodes)
        demand = np.stack([e.get_demand(origin.index, r) for r in range(B)])
        rng = np.random.default_rng(1)
        traj = [obs0.copy()]
        for _ in range(steps):
            obs, rew, *_ = env.step(rng.uniform(0, 4, size=(B, env.n_actions)))
            traj.append(np.concatenate([obs.ravel(), rew.ravel()]))
        return sc.kc.copy(), sc.vf.copy(), sc.od_w.copy(), demand, np.concatenate([t.ravel() for t in traj])

    np.random.seed(0)
    env = VecPedNetEnv("45_intersections", n_envs=B, obs_mode="option3", data_dir=DATA, seed=9)
    a = episode(env, 41)
    b = episode(env, 42)
    c = episode(env, 41)                         # the same env object, reused
    np.random.seed(0)
    env2 = VecPedNetEnv("45_intersections", n_envs=B, obs_mode="option3", data_dir=DATA, seed=9)
    d = episode(env2, 41)                        # a fresh env
    for x, y, z in zip(a, c, d):
        assert np.array_equal(x, y) and np.array_equal(x, z)
    assert not np.array_equal(a[3], b[3]) and not np.array_equal(a[4], b[4])
    env.close()
    env2.close()


def test_multi_scenario_env_groups_have_their_own_topology():
    """randomize_network also moves OD nodes: every group is its own engine.  Group 0 under seed 3 is the scenario of golden
    randnet_i45_a (reference randomize_network('45_intersections', seed=3)); stepping with the widths untouched must
    reproduce that reference run for the env whose RNG key is 0."""
    from pednstream_amd.rl_env import MultiScenarioVecEnv

    g = Golden("randnet_i45_a")
    env = MultiScenarioVecEnv("45_intersections", n_envs=5, group_size=2, obs_mode="option3", data_dir=DATA, seed=0)
    env.reset(options={"randomize": True}, seed=3)
    nets = [grp.network for grp in env.groups]
    assert len(nets) == 3 and [n.n_replicas for n in nets] == [2, 2, 1]
    assert nets[0].origin_nodes == g.meta["origin_nodes"] and nets[0].destination_nodes == g.meta["destination_nodes"]
    assert len({(tuple(n.origin_nodes), tuple(n.destination_nodes), tuple(l.k_critical for l in n._link_list)) for n in nets}) > 1
    steps = 80
    for _ in range(steps):         # t = 1..steps: sending/receiving flows are then defined up to index steps-1
        obs, rew, term, trunc, _ = env.step(np.tile(env.action_high, (5, 1)))      # gates fully open = unchanged widths
    assert obs.shape == (5, env.n_obs) and np.isfinite(obs).all()
    # the groups' launches are enqueued first and fetched afterwards (step_async / step_wait): every group's rows are its own
    off = 0
    for grp in env.groups:
        o, r = grp.network.engine().rl_observe(grp.sim_step - 1, accumulate=False)
        assert np.array_equal(obs[off:off + grp.n_envs], o) and np.array_equal(rew[off:off + grp.n_envs], r)
        off += grp.n_envs
    e = nets[0]._engine
    problems = compare_fields(lambda nm: e.read_block(LINK_FIELDS[nm][0], 0, steps, rep0=0, rep1=1)[:, :, 0].T, g, e.n_links, steps)
    assert not problems, "\n".join(problems)
    env.close()


def test_short_demand_rows_of_a_subset_replace_the_whole_row():
    """ScenarioBatch.commit with demand arrays shorter than T + 1 for SOME replicas: those rows are the array followed by zeros,
    the other replicas keep their demand (one pedn_set_demand_rows upload)."""
    np.random.seed(5)
    net = NetworkEnvGenerator(DATA).create_network("nine_intersections", verbose=False, n_replicas=6, rng_seed=0)
    e = net.engine()
    T = net.simulation_steps
    origin = next(n for n in net.nodes.values() if n.virtual_incoming_link is not None and n.node_id in net.origin_nodes)
    before = e.get_demand(origin.index, 1)
    assert before[200:].sum() > 0
    batch = ScenarioBatch(net)
    short = {0: np.arange(1.0, 31.0), 4: np.full(120, 7.0)}
    for r, arr in short.items():
        batch.set_replica(r, demand={origin.node_id: arr})
    batch.commit()
    for r, arr in short.items():
        got = e.get_demand(origin.index, r)
        assert np.array_equal(got[:len(arr)], arr) and not got[len(arr):].any(), r
    for r in (1, 2, 3, 5):
        assert np.array_equal(e.get_demand(origin.index, r), before), r
    net.run(1, 150)
    model = flatten_network(net)
    for r in (0, 4, 5):              # and the runs follow those rows
        o = od.Oracle(model, seed=0, replica=r)
        o.set_demand(origin.index, short[r] if r in short else before)
        o.run(1, 150)
        for nm in ("cumulative_inflow", "density"):
            mine = e.read_block(LINK_FIELDS[nm][0], 0, 150, rep0=r, rep1=r + 1)[:, :, 0].T
            assert np.array_equal(mine[:e.n_links], o.field(nm)[:e.n_links, :150]), (r, nm)
    net.close()


def test_link_parameter_records_round_trip_and_device_derivation():
    """pedn_set_link_params packs six [L, R] matrices into one 32-byte record per (link, replica) on the device and
    pedn_get_link_params unpacks them: bit for bit what went in.  The randomiser derives travel_time[0] and the two look-backs on
    the device (make_link_pr): equal to scenarios.derive_statics_arrays -- the reference's expressions -- applied to what it drew."""
    from pednstream_amd.scenarios import derive_statics_arrays

    R = 200                                                 # not a multiple of 128: padding lanes exist
    np.random.seed(2)
    net = NetworkEnvGenerator(DATA).create_network("nine_intersections", verbose=False, n_replicas=R, rng_seed=0)
    e = net.engine()
    L = net.n_links
    rng = np.random.default_rng(5)
    kc, vf = rng.uniform(0.5, 3.0, (L, R)), rng.uniform(0.5, 2.0, (L, R))
    kj = kc * rng.uniform(2.0, 4.0, (L, R))
    fft, tsw = rng.integers(0, 3000, (L, R)).astype(np.int32), rng.integers(0, 3000, (L, R)).astype(np.int32)
    tt0 = rng.uniform(1.0, 4000.0, (L, R)).astype(np.float32)
    e.set_link_params(kc, kj, vf, fft, tsw, tt0)
    got = e.get_link_params()
    for name, want in (("kc", kc), ("kj", kj), ("vf", vf), ("fft", fft), ("tau_sw", tsw), ("tt0", tt0)):
        assert got[name].dtype == want.dtype and np.array_equal(got[name], want), name
    with pytest.raises(Exception):
        e.set_link_params(kc, kj, vf, fft + 40000, tsw, tt0)             # beyond the 16 bits of the record
    batch = ScenarioBatch(net)
    batch.draw_random(11)
    length = np.array([l.length for l in net._link_list], dtype=np.float64)[:, None]
    tt0_h, fft_h, tsw_h = derive_statics_arrays(length, batch.vf, batch.kc, batch.kj, net.unit_time)
    assert np.array_equal(batch.tt0, tt0_h) and np.array_equal(batch.fft, fft_h) and np.array_equal(batch.tau_sw, tsw_h)
    net.close()


def test_device_randomiser_draws_the_distributions_of_the_host_generator():
    """pedn_randomize_scenarios against ScenarioBatch.draw_random_host (numpy, the distributions of env_loader.py:183-259,363-424):
    exactly k corridors per env can change and every corridor is equally likely; factor means and variances; OD weights; demand
    patterns and levels; per-replica OD tables equal to the host derivation from the same weights."""
    B = 2048
    np.random.seed(4)
    net = NetworkEnvGenerator(DATA).create_network("45_intersections", verbose=False, n_replicas=B, rng_seed=0)
    e = net.engine()
    dev, host = ScenarioBatch(net), ScenarioBatch(net)
    dev.draw_random(123)
    P = len(dev._pair_links)
    k = int(P * 0.2)
    kc0, kj0, vf0 = (dev._base[n][:, None] for n in ("kc", "kj", "vf"))
    pair_of_link = np.empty(net.n_links, dtype=np.int64)
    for p, links in enumerate(dev._pair_links.values()):
        for l in links:
            pair_of_link[l.index] = p

    def summary(b):
        dens, spd = b.kc != kc0, b.vf != vf0
        assert np.array_equal(dens, b.kj != kj0) or np.all((b.kj != kj0) <= dens)
        touched = np.zeros((P, B), dtype=bool)
        np.logical_or.at(touched, pair_of_link, dens | spd)
        # both directions of a corridor change together
        for links in b._pair_links.values():
            i, j = links[0].index, links[1].index
            assert np.array_equal(dens[i], dens[j]) and np.array_equal(spd[i], spd[j])
        f = (b.kc / kc0)[dens]
        g = (b.vf / vf0)[spd]
        return touched, f, g

    t_d, f_d, g_d = summary(dev)
    assert t_d.sum(axis=0).max() <= k and abs(t_d.sum(axis=0).mean() - 0.75 * k) < 0.15          # P(changes | chosen) = 3/4
    per_corridor = t_d.sum(axis=1) / B                                                          # every corridor: 0.75 k / P
    assert abs(per_corridor.mean() - 0.75 * k / P) < 0.004 and per_corridor.std() < 3.5 * np.sqrt(0.75 * k / P / B)
    assert 0.6 <= f_d.min() and f_d.max() <= 1.2 and abs(f_d.mean() - 0.9) < 0.004 and abs(f_d.var() - 0.03) < 0.002   # U(0.6, 1.2)
    assert 0.6 <= g_d.min() and g_d.max() <= 0.9 + 1e-12 and abs(g_d.mean() - 0.75) < 0.0035 and abs(g_d.var() - 0.0075) < 0.0005
    np.random.seed(4)
    host.draw_random_host(123)
    t_h, f_h, g_h = summary(host)
    assert abs(t_d.mean() - t_h.mean()) < 0.003 and abs(f_d.mean() - f_h.mean()) < 0.006 and abs(g_d.mean() - g_h.mean()) < 0.003
    w = dev.od_w
    n_w = w.size                                                             # U(1, 10): mean 5.5, variance 6.75, 4th central moment 82.0
    assert w.shape == host.od_w.shape and 1.0 <= w.min() and w.max() < 10.0
    assert abs(w.mean() - 5.5) < 4.5 * np.sqrt(6.75 / n_w) and abs(w.var() - 6.75) < 4.5 * np.sqrt((82.0 - 6.75 ** 2) / n_w)
    assert len({w[:, r].tobytes() for r in range(B)}) == B
    # demand: a third of the (origin, env) pairs constant, levels inside the randomiser's ranges
    T = net.simulation_steps
    origin = next(n for n in net.nodes.values() if n.virtual_incoming_link is not None and n.node_id in net.origin_nodes)
    rows = np.stack([e.get_demand(origin.index, r) for r in range(0, B, 4)])
    const = np.all(rows[:, :T] == rows[:, :1], axis=1) & (rows[:, 0] != np.floor(rows[:, 0]))
    assert abs(const.mean() - 1 / 3) < 0.07 and np.all((rows[const, 0] >= 2.0) & (rows[const, 0] < 10.0))
    assert rows.min() >= 0 and 4.0 < rows[~const, :T].mean() < 14.0
    # the tables of P(od | up) derived on the device == the host path fed with the same weights (bit for bit: same operations)
    net.reset(); net.run(1, 12)
    hist_dev = {f: e.read_block(LINK_FIELDS[f][0], 0, 12, rep0=0, rep1=64) for f in ("inflow", "density")}
    tfd = [np.concatenate([e.get_turning_fractions(nd.index, r) for nd in net.nodes.values()]) for r in (0, 1, B - 1)]
    e.set_od_weights_per_replica(w)
    net.reset(); net.run(1, 12)
    for f in hist_dev:
        assert np.array_equal(hist_dev[f], e.read_block(LINK_FIELDS[f][0], 0, 12, rep0=0, rep1=64)), f
    for a, b in zip(tfd, [np.concatenate([e.get_turning_fractions(nd.index, r) for nd in net.nodes.values()]) for r in (0, 1, B - 1)]):
        assert np.array_equal(a, b)
    net.close()


def test_unseeded_vectorised_resets_draw_fresh_scenarios():
    """reset(options={'randomize': True, 'mode': 'vectorised'}) WITHOUT a seed (the usual Gym pattern seeds the first reset
    only) must not draw the same scenario every episode: link parameters, OD weights and demand all differ between two such resets,
    and the seed each one used is kept on the batch."""
    from pednstream_amd.rl_env import VecPedNetEnv

    B = 32
    env = VecPedNetEnv("45_intersections", n_envs=B, obs_mode="option3", data_dir=DATA, seed=9)
    origin = next(n for n in env.network.nodes.values() if n.virtual_incoming_link is not None and n.node_id in env.network.origin_nodes)

    def draw():
        env.reset(options={"randomize": True, "mode": "vectorised"})
        sc, e = env.scenarios, env.network.engine()
        return sc.last_seed, sc.kc.copy(), sc.vf.copy(), sc.od_w.copy(), np.stack([e.get_demand(origin.index, r) for r in range(B)])

    a, b = draw(), draw()
    assert a[0] != b[0]
    for x, y in zip(a[1:], b[1:]):
        assert not np.array_equal(x, y)
    env.reset(options={"randomize": True, "mode": "vectorised"}, seed=a[0])      # and the kept seed reproduces the draw
    assert np.array_equal(env.scenarios.kc, a[1]) and np.array_equal(env.scenarios.od_w, a[3])
    env.close()


def test_device_randomiser_equals_its_cpu_restatement_bit_for_bit():
    """What `mode='vectorised'` draws is the device's own contract (not numpy's stream): oracle/rand_contract.py restates it -- corridor
    selection, factors and floors, OD weights, demand pattern / levels / spike, the Poisson series with glibc's exp -- and the kernels
    must agree with it BIT FOR BIT for every replica checked, whatever the shard (global replica ids key the draws); the same for
    pedn_draw_demand with explicit parameters."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(DATA)), "oracle"))
    import rand_contract as rc
    from pednstream_amd.rl_env import VecPedNetEnv

    B, off, seed = 24, 1000, 0x1234567890ABCDEF
    env = VecPedNetEnv("45_intersections", n_envs=B, obs_mode="option3", data_dir=DATA, seed=9, replica_offset=off)
    env.reset(options={"randomize": True, "mode": "vectorised"}, seed=seed)
    net, sc, e = env.network, env.scenarios, env.network.engine()
    m = e.model
    L, T = e.n_links, int(m["T"])
    rev = np.asarray(m["link_rev"])
    corridors = [(l, int(rev[l])) for l in range(L) if l < rev[l]]
    base = {l: (float(m["link_kc"][l]), float(m["link_kj"][l]), float(m["link_vf"][l])) for l in range(L)}
    kc, kj, vf, od_w = sc.kc, sc.kj, sc.vf, sc.od_w
    origins = [n for n in net.nodes.values() if n.virtual_incoming_link is not None and n.node_id in net.origin_nodes]
    changed = 0
    for r in (0, 1, 7, B - 1):
        want = rc.link_params(seed, off + r, corridors, base)
        for l in range(L):
            assert (kc[l, r], kj[l, r], vf[l, r]) == want[l], (r, l)
        changed += sum(1 for l in range(L) if want[l] != base[l])
        for od in range(od_w.shape[0]):
            assert od_w[od, r] == rc.od_weight(seed, off + r, od), (r, od)
        for node in origins:
            pars = rc.demand_params(seed, off + r, node.index, T)
            series = rc.demand_series(seed, off + r, node.index, T, *pars)
            assert np.array_equal(e.get_demand(node.index, r), np.array(series)), (r, node.node_id, pars)
    assert changed > 0
    # pedn_draw_demand with explicit per-replica parameters: the same series function
    rng = np.random.default_rng(2)
    pattern = rng.integers(0, 3, B).astype(np.int32)
    lo, pk = rng.uniform(2, 10, B), rng.uniform(10, 30, B)
    st, ln, ht = rng.integers(0, T - 30, B).astype(np.int32), rng.integers(10, 20, B).astype(np.int32), rng.integers(20, 50, B).astype(np.float64)
    node = origins[0]
    e.draw_demand(node.index, 77, pattern, lo, pk, st, ln, ht)
    for r in (0, 5, B - 1):
        series = rc.demand_series(77, off + r, node.index, T, int(pattern[r]), float(lo[r]), float(pk[r]), int(st[r]), int(ln[r]), float(ht[r]))
        assert np.array_equal(e.get_demand(node.index, r), np.array(series)), r
    env.close()
