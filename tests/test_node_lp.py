"""assign_flows_type = 'optimal' (node.py:249-271): the node LP.  The reference hands it to scipy/HiGHS; this repository's
oracle and HIP engine solve it with one dense primal simplex (Bland's rule) whose operations are the same on both sides.
The optimum is degenerate -- HiGHS's vertex is not reproducible -- so the pin is objective-level: same optimal value, every
constraint satisfied.  PARITY UNPINNED beyond that."""
import ctypes as C

import numpy as np
import pytest

import oracle_driver as od

scipy_opt = pytest.importorskip("scipy.optimize")


def reference_lp(m, tf):
    """A_ub / A_eq / c exactly as Node.get_matrix_A (:73-98), update_matrix_A_eq (:110-137) and RegularNode.solve (:249-254) build them."""
    E = m * (m - 1)
    A_ub = np.zeros((2 * m, m * m + 2 * E))
    for i in range(m):
        e = np.ones(m)
        e[i] = 0
        A_ub[i, i * m:(i + 1) * m] = e
    for j in range(m):
        for k in range(m):
            A_ub[m + j, j + k * m] = 1 if k != j else 0
    A_ub = np.delete(A_ub, [i * m + i for i in range(m)], axis=1)
    A_eq = np.zeros((E, 3 * E))
    for i in range(E):
        src = i // (m - 1)
        A_eq[i, src * (m - 1):(src + 1) * (m - 1)] = tf[i]
        A_eq[i, i] = tf[i] - 1
        A_eq[i, E + 2 * i:E + 2 * i + 2] = [1, -1]
    c = np.concatenate([-np.ones(E), 1e-2 * np.ones(2 * E)])
    return A_ub, A_eq, c


def oracle_lp(m, s, r, tf):
    L = od.lib()
    L.pedn_oracle_lp.restype = C.c_int
    P = C.POINTER(C.c_double)
    E = m * (m - 1)
    x, g = np.zeros(3 * E + 2 * m), np.zeros(E)
    s, r, tf = (np.ascontiguousarray(a, dtype=np.float64) for a in (s, r, tf))
    rc = L.pedn_oracle_lp(m, s.ctypes.data_as(P), r.ctypes.data_as(P), tf.ctypes.data_as(P), x.ctypes.data_as(P), g.ctypes.data_as(P))
    return rc, x, g


def test_simplex_reaches_the_objective_of_scipy_highs_on_random_nodes():
    rng = np.random.default_rng(0)
    for trial in range(250):
        m = int(rng.integers(3, 8))
        E = m * (m - 1)
        tf = rng.random((m, m - 1))
        tf = (tf / tf.sum(axis=1, keepdims=True)).ravel()
        if trial % 5 == 0:
            tf = np.full(E, 1 / (m - 1))                                   # the default fractions (network.py:269-271)
        s = np.floor(rng.random(m) * rng.choice([0, 5, 40, 200], m))
        r = np.floor(rng.random(m) * rng.choice([0, 10, 60, 1e6], m))
        A_ub, A_eq, c = reference_lp(m, tf)
        res = scipy_opt.linprog(c, A_ub=A_ub, A_eq=A_eq, b_ub=np.concatenate([s, r]), b_eq=np.zeros(E))
        rc, x, g = oracle_lp(m, s, r, tf)
        assert rc == 0 and res.success
        xs = x[:3 * E]
        assert abs(c @ xs - res.fun) <= 1e-8 * max(1.0, abs(res.fun)), (trial, c @ xs, res.fun)
        assert (A_ub @ xs <= np.concatenate([s, r]) + 1e-7).all() and np.abs(A_eq @ xs).max() <= 1e-7 and xs.min() >= -1e-9
        assert np.array_equal(g, np.floor(x[:E]))
        # what the reference then does with it: q = max(0, A_ub @ floor(x)) -- total throughput agrees to within the floors
        q_ref, q_mine = A_ub @ np.floor(res.x), A_ub[:, :E] @ g
        assert abs(q_ref[:m].sum() - q_mine[:m].sum()) <= E


@pytest.mark.parametrize("case", ["lp_nine", "lp_i45"])
def test_simplex_reaches_the_objective_of_the_reference_on_its_own_solves(case):
    """tests/golden/lp_*.npz: every programme the REAL reference solved (scipy/HiGHS inside RegularNode.solve, node.py:263-266)
    in the first steps of a run with assign_flows_type 'optimal' -- inputs, optimal value, resulting q.  The oracle's simplex
    must reach each optimal value; the throughput q it yields may differ by the floors of a different vertex only."""
    import os

    from golden_util import GOLDEN

    z = np.load(os.path.join(GOLDEN, case + ".npz"))
    assert z["success"].all()
    n_same_q = 0
    for k in range(len(z["m"])):
        m = int(z["m"][k])
        E = m * (m - 1)
        s, r, tf = z["s"][k, :m], z["r"][k, :m], z["tf"][k, :E]
        rc, x, g = oracle_lp(m, s, r, tf)
        assert rc == 0
        _, _, c = reference_lp(m, tf)
        assert abs(c @ x[:3 * E] - z["fun"][k]) <= 1e-8 * max(1.0, abs(z["fun"][k])), (k, c @ x[:3 * E], z["fun"][k])
        A_ub = reference_lp(m, tf)[0][:, :E]
        q = np.maximum(0, A_ub @ g)
        assert abs(q[:m].sum() - z["q"][k, :m].sum()) <= E
        n_same_q += np.array_equal(q, z["q"][k, :2 * m])
    assert n_same_q >= 0.5 * len(z["m"])      # most programmes have a unique optimum: then even q is the reference's


def test_oracle_runs_a_scenario_with_the_lp_node_model():
    """forky (static fractions) with assign_flows_type 'optimal': no error flag, flows conserved at every node and never above
    the sending / receiving flows -- the constraints of the LP -- and different from the classic model somewhere."""
    from golden_util import Golden
    from pednstream_amd import Network
    from pednstream_amd.flatten import flatten_network

    g = Golden("forky")
    outs = {}
    for kind in ("classic", "optimal"):
        np.random.seed(g.info["np_seed"])
        net = Network(np.array(g.info["adjacency"]), dict(g.info["params"], assign_flows_type=kind), origin_nodes=g.info["origin_nodes"],
                      destination_nodes=g.info["destination_nodes"], verbose=False)
        if g.info["tf_nodes"]:
            net.update_turning_fractions_per_node(g.info["tf_nodes"], np.array(g.info["tf_values"]))
        for nid, arr in g.demand().items():
            net.nodes[nid].demand = arr
        model = flatten_network(net)
        assert model["node_model"] == (1 if kind == "optimal" else 0)
        o = od.Oracle(model, seed=g.seed, replica=g.replica)
        for node in net.nodes.values():
            tf = net._tf_host.get(node.index)
            if tf is not None:
                o.set_tf(node.index, tf[:, 0])
        o.run(1, g.steps)
        assert o.flags() == 0
        outs[kind] = {f: o.field(f) for f in ("inflow", "outflow", "sending_flow", "receiving_flow", "cumulative_inflow")}
    L = model["n_links"]
    opt = outs["optimal"]
    assert (opt["outflow"][:L, 1:g.steps] <= opt["sending_flow"][:L, :g.steps - 1] + 1e-9).all()      # out[t] <= S[t-1]
    assert (opt["inflow"][:L, 1:g.steps] <= opt["receiving_flow"][:L, :g.steps - 1] + 1e-9).all()
    assert opt["cumulative_inflow"][:L, g.steps - 1].sum() > 0
    assert not np.array_equal(opt["inflow"], outs["classic"]["inflow"])


@pytest.mark.gpu
@pytest.mark.parametrize("scenario,steps", [("nine_intersections", 80), ("od_flow_example", 150), ("long_corridor", 120)])
def test_engine_equals_oracle_with_the_lp_node_model(scenario, steps):
    """Same simplex, same operations: the HIP engine and the CPU oracle agree bit for bit under assign_flows_type 'optimal'
    (dynamic turning fractions, separators, several replicas)."""
    from golden_util import ALL_FIELDS, DATA
    from pednstream_amd import NetworkEnvGenerator
    from pednstream_amd.flatten import flatten_network
    from pednstream_amd.network import LINK_FIELDS

    np.random.seed(3)
    gen = NetworkEnvGenerator(DATA)
    gen.network_data = gen.load_network_data(scenario)
    gen.config["params"]["assign_flows_type"] = "optimal"
    net = gen.create_network(scenario, verbose=False, n_replicas=70, rng_seed=9)
    assert net.assign_flows_type == "optimal"
    model = flatten_network(net)
    net.run(1, steps)
    e = net._engine
    for r in (0, 37, 69):
        o = od.Oracle(model, seed=9, replica=r)
        o.run(1, steps)
        assert o.flags() == 0
        for name in ALL_FIELDS:
            mine = e.read_block(LINK_FIELDS[name][0], 0, steps, rep0=r, rep1=r + 1)[:, :, 0].T
            assert np.array_equal(mine[:e.n_links], o.field(name)[:e.n_links, :steps]), (r, name)
        o.close()
    assert e.read_block(2, steps - 1, steps).sum() > 0
    net.close()
