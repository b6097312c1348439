"""The reference's raise sites become sticky per-replica flags (include/pedn.h PEDN_F_*): oracle on CPU, engine on GPU."""
import numpy as np
import pytest

import oracle_driver as od
from pednstream_amd import Network
from pednstream_amd.flatten import flatten_network

ADJ = np.array([[0, 1, 0], [1, 0, 1], [0, 1, 0]])


def params(length):
    return {"unit_time": 10, "simulation_steps": 60, "assign_flows_type": "classic", "seed": 1,
            "default_link": {"length": length, "width": 2, "free_flow_speed": 1.1, "k_critical": 2, "k_jam": 6, "gamma": 0.01},
            "demand": {"origin_0": {"peak_lambda": 10, "base_lambda": 8}}}


def test_zero_step_lookback_is_flagged_by_the_oracle():
    """A 4 m link is crossed in 0.36 time steps: tau = round(avg_travel_time / dt) = 0 and the sending flow would read
    cumulative_inflow[t], which another node writes in the same step -- order dependent in the reference."""
    net = Network(ADJ, params(4), origin_nodes=[0], verbose=False)
    assert net.links[(0, 1)].free_flow_tau == 0
    o = od.Oracle(flatten_network(net))
    o.run(1, 30)
    assert o.flags() & 16
    ok = od.Oracle(flatten_network(Network(ADJ, params(40), origin_nodes=[0], verbose=False)))
    ok.run(1, 30)
    assert ok.flags() == 0


def test_negative_demand_is_flagged_by_the_oracle():
    net = Network(ADJ, params(40), origin_nodes=[0], verbose=False)
    net.nodes[0].demand[10] = -3
    o = od.Oracle(flatten_network(net))
    o.run(1, 30)
    assert o.flags() & 2          # node.py:218-219 "Negative flows detected"


@pytest.mark.gpu
def test_engine_raises_like_the_reference():
    from pednstream_amd import engine as eng

    net = Network(ADJ, params(4), origin_nodes=[0], verbose=False, n_replicas=5)
    # with free_flow_tau == 0 the reference itself raises ValueError("Negative sending flow ...") at t' = 0
    # (floor(0.8*0 + 0.2*sending_flow[-1]) = -1, link.py:364-366); the order-dependence flag is set as well
    with pytest.raises(ValueError) as ei:
        net.run(1, 30)
    assert (ei.value.flags & 16).all() and (ei.value.flags & 1).all()
    net.close()
    net = Network(ADJ, params(40), origin_nodes=[0], verbose=False, n_replicas=3)
    net.nodes[0].demand[10] = -3
    net.run(1, 8)                                   # nothing wrong yet
    with pytest.raises(eng.ModelError) as ei:
        net.run(8, 30)
    assert (ei.value.flags & 2).all()
    # degree above the kernel limit is rejected at creation with a clear message
    big = np.zeros((11, 11), dtype=int)
    big[0, 1:] = big[1:, 0] = 1
    hub = Network(big, params(40), origin_nodes=[1], verbose=False)
    with pytest.raises(RuntimeError, match="degree"):
        hub.engine()
    net.close()
