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
    e = nets[0]._engine
    problems = compare_fields(lambda nm: e.read_block(LINK_FIELDS[nm][0], 0, steps, rep0=0, rep1=1)[:, :, 0].T, g, e.n_links, steps)
    assert not problems, "\n".join(problems)
    env.close()
