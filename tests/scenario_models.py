"""Oracle models of single replicas of a batch with per-replica scenarios -- TEST INFRASTRUCTURE.

`ScenarioBatch` gives every replica of ONE engine its own k_critical / k_jam / free-flow speed (+ the look-backs derived from
them), time-constant OD weights and origin demand.  The CPU oracle runs one replica at a time, so the checker of replica r is
an oracle built from the shared model with replica r's columns put in place."""
import numpy as np


def replica_model(model, net, batch, r, demand_from_engine=True):
    """The flattened model of replica `r` of `net` under `batch` (a committed ScenarioBatch).  The demand rows come from the
    engine itself (`pedn_get_demand`), because a committed batch no longer holds them (and device-drawn series never were
    on the host)."""
    T = int(model["T"])
    mr = dict(model)
    mr["link_kc"], mr["link_kj"], mr["link_vf"] = batch.kc[:, r].copy(), batch.kj[:, r].copy(), batch.vf[:, r].copy()
    mr["link_fft"], mr["link_tau_sw"], mr["link_tt0"] = batch.fft[:, r].copy(), batch.tau_sw[:, r].copy(), batch.tt0[:, r].copy()
    if batch.od_w is not None:
        mr["od_w"] = np.repeat(batch.od_w[:, r:r + 1], T + 1, axis=1)
    if demand_from_engine:
        e = net.engine()
        d = np.array(model["demand"], dtype=np.float64).copy()
        for node in net.nodes.values():
            row = model["node_demand_row"][node.index]
            if row >= 0:
                d[row, :] = e.get_demand(node.index, r)[:d.shape[1]]
        mr["demand"] = d
    return mr
