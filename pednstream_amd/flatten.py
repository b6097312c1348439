"""Freeze a host-side ``Network`` into the flat CSR arrays of the C-ABI model description (include/pedn.h).

Everything the device needs is replica independent and static: node -> incident-link slots (incoming[k] and
outgoing[k] are the two directions towards the same neighbour, virtual pair at slot 0 -- network.py:125-128,
236-240), per-link parameters, demand rows, OD weights and the route-choice tables of
``pednstream_amd.path_finder`` with their Python iteration orders turned into array orders.

Static per-link scalars whose defining expression is pure Python arithmetic in the reference are evaluated
here with that same expression (e.g. ``round(length / (shockwave_speed * unit_time))``, link.py:380), so that
the device never has to re-derive them with possibly different rounding.
"""
import numpy as np

from .network import FD_TYPES

I32, F64, F32 = np.int32, np.float64, np.float32


def flatten_network(net) -> dict:
    nodes = list(net.nodes.values())
    links = net._link_list
    L, N = len(links), len(nodes)
    T = int(net.simulation_steps)
    m = {"n_nodes": N, "n_links": L, "n_vlinks": int(net.n_vlinks), "T": T, "dt": float(net.unit_time),
         "history_mode": {"full": 0, "recent": 1}[getattr(net, "history", "full")],
         "node_model": {"classic": 0, "optimal": 1}[getattr(net, "assign_flows_type", "classic")]}

    # ---- nodes / slots -------------------------------------------------------------------------------------
    slot_ptr, turn_ptr = [0], [0]
    slot_in, slot_out = [], []
    kind, dem_row = [], []
    demand_rows = []
    for nd in nodes:
        assert nd.source_num == nd.dest_num, "incoming/outgoing slot counts must match (bidirectional links)"
        for k in range(nd.source_num):
            lin, lout = nd.incoming_links[k], nd.outgoing_links[k]
            if not lin.is_virtual:
                assert lin.reverse_link is lout, "slot k must hold the two directions of one corridor"
            slot_in.append(lin.index)
            slot_out.append(lout.index)
        slot_ptr.append(len(slot_in))
        turn_ptr.append(turn_ptr[-1] + nd.edge_num)
        kind.append(0 if nd.kind == "one_to_one" else 1)
        if nd.virtual_incoming_link is not None:
            row = np.zeros(T + 1)
            d = np.asarray(nd._demand, dtype=F64)
            row[:min(len(d), T + 1)] = d[:T + 1]
            dem_row.append(len(demand_rows))
            demand_rows.append(row)
        else:
            dem_row.append(-1)
    m["node_id"] = np.array([nd.node_id for nd in nodes], dtype=I32)
    # measured packing cost per node (tools/pack_calibrate.py), when the scenario directory carries one for every node
    cost = getattr(net, "node_pack_cost", None)
    m["node_cost"] = (np.array([cost[nd.node_id] for nd in nodes], dtype=F32) if cost and all(nd.node_id in cost for nd in nodes) else None)
    m["node_kind"] = np.array(kind, dtype=I32)
    m["node_slot_ptr"] = np.array(slot_ptr, dtype=I32)
    m["node_turn_ptr"] = np.array(turn_ptr, dtype=I32)
    m["node_demand_row"] = np.array(dem_row, dtype=I32)
    m["slot_in_link"] = np.array(slot_in, dtype=I32)
    m["slot_out_link"] = np.array(slot_out, dtype=I32)
    m["n_demand"] = len(demand_rows)
    m["demand"] = np.array(demand_rows, dtype=F64).reshape(len(demand_rows), T + 1)
    m["n_turns"] = int(turn_ptr[-1])
    m["max_degree"] = int(max(nd.source_num for nd in nodes))
    tf0 = np.empty(turn_ptr[-1], dtype=F64)
    for nd, a, b in zip(nodes, turn_ptr[:-1], turn_ptr[1:]):
        tf0[a:b] = 1 / (nd.dest_num - 1)          # network.py:269-271
    m["tf_init"] = tf0

    # ---- links ---------------------------------------------------------------------------------------------
    def col(fn, dtype=F64):
        return np.array([fn(l) for l in links], dtype=dtype)

    m["link_u"] = col(lambda l: l.start_node.node_id, I32)
    m["link_v"] = col(lambda l: l.end_node.node_id, I32)
    m["link_rev"] = col(lambda l: l.reverse_link.index, I32)
    m["link_sep"] = col(lambda l: int(l.is_separator), I32)
    m["link_fd"] = col(lambda l: FD_TYPES[l.fd_type], I32)
    m["link_length"] = col(lambda l: l.length)
    m["link_width"] = col(lambda l: l._width)
    m["link_vf"] = col(lambda l: l.free_flow_speed)
    m["link_kc"] = col(lambda l: l.k_critical)
    m["link_kj"] = col(lambda l: l.k_jam)
    m["link_gamma"] = col(lambda l: l.gamma)
    m["link_act"] = col(lambda l: l.activity_probability)
    m["link_bi"] = col(lambda l: l.bi_factor)
    m["link_noise"] = col(lambda l: l.speed_noise_std)
    m["link_tau_sw"] = col(lambda l: l.tau_shockwave, I32)
    m["link_fft"] = col(lambda l: l.free_flow_tau, I32)
    m["link_tt0"] = col(lambda l: l.travel_time0, F32)
    m["window"] = int(links[0].avg_travel_time_window) if links else 1
    m["front_gate0"] = col(lambda l: l._init_widths[0])
    m["back_gate0"] = col(lambda l: l._init_widths[1])
    m["sep_width0"] = col(lambda l: l._init_widths[2])

    # ---- OD weights + route-choice tables ---------------------------------------------------------------
    pf = net.path_finder
    if pf is None:
        m["n_od"] = 0
        m["od_w"] = np.zeros((0, T + 1))
        pfp = (0.0, 0.0, 0.0, 0.0, 0.0)
        od_index = {}
    else:
        od_index = {od: i for i, od in enumerate(net.od_manager.od_flows.keys())}
        m["n_od"] = len(od_index)
        m["od_w"] = net.od_manager.as_matrix()
        pfp = (pf.temp, pf.alpha, pf.beta, pf.omega, pf.epsilon)
    m["pf_temp"], m["pf_alpha"], m["pf_beta"], m["pf_omega"], m["pf_eps"] = (float(x) for x in pfp)

    node_dyn = np.zeros(N, dtype=I32)
    node_up_ptr, up_slot, up_od_ptr, upod_od = [0], [], [0], []
    node_grp_ptr, grp_ent_ptr, grp_allphys, grp_up, grp_node, ent_link, ent_dist = [0], [0], [], [], [], [], []
    pair_ptr = [0]
    pair_ent, pair_upod = [], []
    for nd in nodes:
        dyn = (pf is not None and nd.node_id in pf.nodes_in_paths and nd.source_num > 2
               and nd.node_id in pf.tables)
        if dyn:
            node_dyn[nd.index] = 1
            tbl = pf.tables[nd.node_id]
            up_ids = [(-1 if l.is_virtual else l.start_node.node_id) for l in nd.incoming_links]
            dn_ids = [(-1 if l.is_virtual else l.end_node.node_id) for l in nd.outgoing_links]
            upod_index = {}     # (up id, od) -> global index into upod_od
            for up, ods in tbl.up_od_probs.items():
                up_slot.append(up_ids.index(up))
                for od in ods.keys():
                    upod_index[(up, od)] = len(upod_od)
                    upod_od.append(od_index[od])
                up_od_ptr.append(len(upod_od))
            ent_index = {}      # (od, up id, down id) -> global entry index
            for od, ups in tbl.turns_distances.items():
                for up, downs in ups.items():
                    if not downs:
                        continue
                    allphys = 1
                    for dn, dist in downs.items():
                        ent_index[(od, up, dn)] = len(ent_link)
                        if (nd.node_id, dn) in net.links:
                            ent_link.append(net.links[(nd.node_id, dn)].index)
                        else:
                            ent_link.append(-1)          # KeyError branch of path_finder.py:577-579
                            allphys = 0
                        ent_dist.append(float(dist))
                    grp_ent_ptr.append(len(ent_link))
                    grp_allphys.append(allphys)
                    grp_up.append(up_ids.index(up))
                    grp_node.append(nd.index)
            for i, up in enumerate(up_ids):
                for j, dn in enumerate(dn_ids):
                    if up == dn:
                        continue
                    for od in tbl.ods_in_turns.get((up, dn), set()):
                        pair_ent.append(ent_index[(od, up, dn)])
                        pair_upod.append(upod_index[(up, od)])
                    pair_ptr.append(len(pair_ent))
        else:
            pair_ptr.extend([len(pair_ent)] * nd.edge_num)
        node_up_ptr.append(len(up_slot))
        node_grp_ptr.append(len(grp_allphys))
    assert len(pair_ptr) == m["n_turns"] + 1
    m["node_dyn"] = node_dyn
    m["node_up_ptr"] = np.array(node_up_ptr, dtype=I32)
    m["up_slot"] = np.array(up_slot, dtype=I32)
    m["up_od_ptr"] = np.array(up_od_ptr, dtype=I32)
    m["upod_od"] = np.array(upod_od, dtype=I32)
    m["node_grp_ptr"] = np.array(node_grp_ptr, dtype=I32)
    m["grp_ent_ptr"] = np.array(grp_ent_ptr, dtype=I32)
    m["grp_allphys"] = np.array(grp_allphys, dtype=I32)
    m["grp_up"] = np.array(grp_up, dtype=I32)
    m["grp_node"] = np.array(grp_node, dtype=I32)
    m["ent_link"] = np.array(ent_link, dtype=I32)
    m["ent_dist"] = np.array(ent_dist, dtype=F64)
    m["turn_pair_ptr"] = np.array(pair_ptr, dtype=I32)
    m["pair_ent"] = np.array(pair_ent, dtype=I32)
    m["pair_upod"] = np.array(pair_upod, dtype=I32)
    m["n_up"], m["n_upod"], m["n_grp"], m["n_ent"], m["n_pair"] = (len(up_slot), len(upod_od), len(grp_allphys),
                                                                   len(ent_link), len(pair_ent))
    return m
