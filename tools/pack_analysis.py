#!/usr/bin/env python3
"""Is the wait at node_kernel's first block barrier a property of WHICH slots share a block (then re-packing the bins by measured wave
cost would remove it) or of step-to-step variation (then it would not)?  VERDICT r04 item 6.

Needs the profiling build (`make -C pednstream_amd/csrc phase-profile`) and a GPU:

    PEDN_STREAMS=1 python tools/pack_analysis.py delft melbourne

Per wave of the grid the build accumulates, over the steps run, the ticks between its phase stamps.  From them, per network:
  measured wait      mean ticks a wave spends at barrier 1 (phase 6)
  systematic wait    what the MEAN arrival times alone explain: per block, max over its waves of the mean time from kernel entry to
                     barrier 1, minus each wave's own mean -- the part a static re-packing could address
  ideal re-packing   the same after sorting all slots by their mean arrival time and re-binning them greedily into blocks of eight
                     waves with nodes kept whole (an upper bound on what any static packing gains)"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "pednstream_amd", "csrc", "libpedn_hip_phase.so")
os.environ["PEDN_HIP_LIB"] = LIB
os.environ.setdefault("PEDN_STREAMS", "1")       # one chain: the wave index of the accumulators is the grid's

from bench import replica_demand  # noqa: E402
from pednstream_amd import NetworkEnvGenerator  # noqa: E402


def host_bins(model):
    """pedn_create's packing by the static estimate (nodes by (load, degree) decreasing, first fit into blocks of eight waves), rebuilt
    here to know which waves of a block belong to which node."""
    nsp, ntp, tpp = np.asarray(model["node_slot_ptr"]), np.asarray(model["node_turn_ptr"]), np.asarray(model["turn_pair_ptr"])
    N = len(nsp) - 1
    deg = np.diff(nsp)
    load = tpp[ntp[1:]] - tpp[ntp[:-1]]
    order = sorted(range(N), key=lambda n: (-load[n], -deg[n]))                # Python's sort is stable, like std::stable_sort
    return first_fit(order, deg), deg


def first_fit(order, deg):
    bins, fill = [], []
    for n in order:
        for b in range(len(bins)):
            if fill[b] + deg[n] <= 8:
                break
        else:
            bins.append([])
            fill.append(0)
            b = len(bins) - 1
        bins[b].append(n)
        fill[b] += deg[n]
    return bins


def measure(lib, network, R=1024, steps=200):
    """(net, per-wave means [bin][replica group][wave] of: ticks from kernel entry to barrier 1, wait at barrier 1, lifetime) under the
    static packing, one chain, with the bench's per-replica demand."""
    os.environ["PEDN_PACK_BY_LOAD"] = "1"          # the packing host_bins() rebuilds, whatever the scenario directory holds
    net = NetworkEnvGenerator(os.path.join(ROOT, "data")).create_network(network, verbose=False, n_replicas=R, rng_seed=0)
    e = net.engine()
    for nid in net.origin_nodes:
        e.set_demand_matrix(net.nodes[nid].index, np.stack([replica_demand(net.simulation_steps, r) for r in range(R)]))
    e.run(1, 150)
    e.synchronize()
    lib.pedn_debug_phases(None, 1)
    e.run(150, 150 + steps)
    e.synchronize()
    groups = R // 64
    n_waves = 1 << 17
    raw = (ctypes.c_ulonglong * (n_waves * 12))()
    assert lib.pedn_debug_phase_waves(raw, n_waves) == 0
    a = np.frombuffer(raw, dtype=np.uint64).reshape(n_waves, 12).astype(np.float64)
    used = np.flatnonzero(a[:, 10] > 0)
    n_blocks = (used.max() // 8 + 1) // groups
    a = a[:n_blocks * groups * 8].reshape(n_blocks, groups, 8, 12)          # [bin][replica group][wave][phase]
    cnt = a[..., 10]
    live = cnt > 0
    arrive = np.where(live, a[..., 1:6].sum(axis=-1) / np.maximum(cnt, 1), np.nan)     # mean ticks from entry to barrier 1
    wait = np.where(live, a[..., 6] / np.maximum(cnt, 1), np.nan)
    life = np.where(live, a[..., 11] / np.maximum(cnt, 1), np.nan)
    return net, arrive, wait, life


def node_costs(model, arrive):
    """node index -> (cost of its slowest slot wave, the costs of its slots), from the per-wave means."""
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore", category=RuntimeWarning)          # idle waves of the last block: all-NaN slices
        slot_cost = np.nanmean(arrive, axis=1)                             # [bin][wave], averaged over the replica groups
    bins, deg = host_bins(model)
    assert len(bins) == arrive.shape[0], (len(bins), arrive.shape)
    cost, slots = {}, {}
    for b, nodes in enumerate(bins):
        w = 0
        for n in nodes:
            c = slot_cost[b, w:w + deg[n]]
            cost[n], slots[n] = float(np.nanmax(c)), c
            w += deg[n]
    return bins, deg, slot_cost, cost, slots


def main():
    lib = ctypes.CDLL(LIB)
    for network in sys.argv[1:] or ["delft"]:
        R = 1024
        net, arrive, wait, life = measure(lib, network, R)
        e = net.engine()
        groups, n_blocks = R // 64, arrive.shape[0]
        measured = np.nanmean(wait)
        systematic = np.nanmean(np.nanmax(arrive, axis=2, keepdims=True) - arrive)
        bins, deg, slot_cost, node_cost, node_slots = node_costs(e.model, arrive)
        N = len(deg)
        own, others = [], []
        for nodes in bins:
            bmax = max(node_cost[n] for n in nodes)
            for n in nodes:
                own.extend(node_cost[n] - node_slots[n])
                others.extend([bmax - node_cost[n]] * deg[n])
        # ideal node-respecting static packing: nodes by their measured cost, decreasing, first fit
        bins2 = first_fit(sorted(range(N), key=lambda n: -node_cost[n]), deg)
        others2 = []
        for nodes in bins2:
            bmax = max(node_cost[n] for n in nodes)
            for n in nodes:
                others2.extend([bmax - node_cost[n]] * deg[n])
        lifem = np.nanmean(life)
        print(f"== {network} x {R}, {n_blocks} blocks x {groups} replica groups, 200 steps: mean wave lifetime {lifem:.0f} ticks")
        print(f"   wait at barrier 1, measured (mean over waves and steps)                  {measured:7.0f} ticks  {100 * measured / lifem:5.1f} % of a wave's life")
        print(f"   explained by the waves' MEAN arrival times (static, per wave)            {systematic:7.0f} ticks")
        print(f"   of which a wave waits for the slower waves of its OWN node (inherent)    {np.mean(own):7.0f} ticks")
        print(f"   ... and for the slowest OTHER node of its block (the packing's)          {np.mean(others):7.0f} ticks")
        print(f"   the same under an ideal static packing by measured node cost ({len(bins2)} blocks)  {np.mean(others2):7.0f} ticks  -> a static re-packing "
              f"can gain at most {np.mean(others) - np.mean(others2):.0f} ticks = {100 * (np.mean(others) - np.mean(others2)) / lifem:.1f} % of a wave's life")
        print(f"   spread of slot costs: min {np.nanmin(slot_cost):.0f}  median {np.nanmedian(slot_cost):.0f}  max {np.nanmax(slot_cost):.0f} ticks")
        net.close()


if __name__ == "__main__":
    main()
