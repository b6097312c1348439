"""The CPU oracle (oracle/pedn_oracle.c) pinned against the real reference: every golden captured from
/root/reference under the injected RNG must be reproduced bit for bit on all 13 per-link arrays."""
import numpy as np
import pytest

from golden_util import DIGEST_CASES, Golden, compare_digests, compare_fields, run_oracle

CASES = ["six_node_full", "nine_full", "long_corridor_full", "small_network_full", "i45_prefix", "delft_prefix",
         "melbourne_prefix", "nine_meanfield", "six_node_gate", "forky", "nine_replica0", "nine_replica1",
         "nine_replica2", "nine_replica3", "odd_params", "odd_separators", "star8", "edge_window_gt_T", "edge_empty", "rand_nine_a", "rand_nine_b", "rand_delft_a",
         "rand_delft_b", "randnet_i45_a", "randnet_i45_b", "randnet_nine", "butterfly_scA_full", "butterfly_scB_full", "butterfly_scC_full",
         "one_intersection_full", "two_coordinators_prefix", "spike_callable", "melbourne_callable", "forky_front", "edge_T2", "edge_T3"]


@pytest.mark.parametrize("case", CASES)
def test_oracle_reproduces_reference_bit_exact(case):
    g = Golden(case)
    o, tfh, model, net = run_oracle(g)
    assert o.flags() == 0
    problems = compare_fields(o.field, g, model["n_links"], g.steps)
    assert not problems, "\n".join(problems)
    # virtual links' flow arrays (update_links writes them too, node.py:154-161)
    L = model["n_links"]
    if model["n_vlinks"]:
        for tag, off in (("vin", 0), ("vout", 1)):
            for name in ("inflow", "outflow", "cumulative_inflow", "cumulative_outflow"):
                mine = o.field(name)[L + off::2, :g.steps]
                assert np.array_equal(mine, g.state(f"{tag}_{name}")[:, :g.steps]), (tag, name)
    # turning fractions: the softmax's exp() is glibc 2.35's, restated with its fused multiply-adds -> bit-exact too
    ref_tf = g.z["tf_hist"]
    assert tfh.shape == ref_tf.shape
    assert np.array_equal(tfh, ref_tf)


@pytest.mark.parametrize("case", DIGEST_CASES)
def test_oracle_reproduces_full_horizon_reference_runs(case):
    """melbourne / delft over ALL 499 steps (and melbourne under heavy demand, where the release binomials and the diffusion
    look-backs fire): every array at every step through per-step digests over all links + complete arrays of 16 links."""
    g = Golden(case)
    o, tfh, model, net = run_oracle(g)
    assert o.flags() == 0 and g.steps == model["T"]
    problems = compare_digests(o.field, g, model["n_links"], tfh)
    assert not problems, "\n".join(problems)
    assert o.field("cumulative_inflow")[:model["n_links"], g.steps - 1].sum() == g.info["totals"]["cumulative_inflow_last"]
