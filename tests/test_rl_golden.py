"""RL glue (config #5): agent discovery, action clipping, observations and rewards against goldens captured from the
reference's own rl/discovery.py, rl/builders.py and the reward function of rl/pz_pednet_env.py."""
import numpy as np
import pytest

from golden_util import Golden, build_network, compare_fields
from pednstream_amd.flatten import flatten_network
from pednstream_amd.rl_env import AgentManager
from rl_oracle import RlOracle

RL_CASES = ["rl_nine_opt3", "rl_nine_opt2n", "rl_nine_opt5g2", "rl_nine_opt4", "rl_i45_opt3", "rl_corridor_opt1", "rl_butterfly_opt3",
            "rl_one_intersection_opt5", "rl_nine_partial", "rl_corridor_partial", "rl_i45_episode"]


def agent_spec(net):
    am = AgentManager(net)
    spec = []
    for aid in am.get_all_agent_ids():
        if am.get_agent_type(aid) == "sep":
            f, r = am.get_separator_links(aid)
            spec.append({"id": aid, "type": "sep", "links": [f.link_id, r.link_id]})
        else:
            spec.append({"id": aid, "type": "gate", "links": [l.link_id for l in am.get_gater_outgoing_links(aid)]})
    return spec


@pytest.mark.parametrize("case", RL_CASES)
def test_agent_discovery_matches_reference(case):
    g = Golden(case)
    assert agent_spec(build_network(g)) == g.info["rl"]["agents"]


@pytest.mark.parametrize("case", RL_CASES)
def test_rl_restatement_on_cpu_oracle_matches_reference(case):
    g = Golden(case)
    rl = g.info["rl"]
    net = build_network(g)
    model = flatten_network(net)
    env = RlOracle(net, model, rl["agents"], rl["obs_mode"], rl["normalize"], rl["action_gap"], seed=g.seed, replica=g.replica)
    acts, ref_obs, ref_rew = g.state("rl_actions"), g.state("rl_obs"), g.state("rl_rewards")
    for k in range(rl["env_steps"]):
        obs, rew = env.step(acts[k])
        assert np.array_equal(obs, ref_obs[k]), (k, obs, ref_obs[k])
        assert np.array_equal(rew, ref_rew[k]), (k, rew, ref_rew[k])
    assert not compare_fields(env.o.field, g, model["n_links"], g.steps)
    assert not g.state("rl_terminated").any()
