"""Random small scenarios for fuzz tests (same generator as oracle/fuzz_vs_reference.py, which ran them through the real
reference offline; here they only need this repository's own host code)."""
import numpy as np


def random_case(seed, short_links=False):
    """short_links: about a third of the corridors are shorter than half a time step (tau = 0 and / or tau_shockwave = 0):
    their look-backs reach into the step that is being computed, where the reference's result depends on its node order."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(5, 14))
    adj = np.zeros((n, n), dtype=int)
    for i in range(1, n):
        j = int(rng.integers(0, i))
        adj[i, j] = adj[j, i] = 1
    for _ in range(int(rng.integers(0, n))):
        i, j = rng.integers(0, n, size=2)
        if i != j and adj[i].sum() < 6 and adj[j].sum() < 6:
            adj[i, j] = adj[j, i] = 1
    dt = float(rng.choice([5, 10, 10, 20]))
    T = int(rng.integers(80, 160))
    fd = str(rng.choice(["yperman", "greenshields", "smulders"]))
    default = {"length": float(rng.uniform(35, 120)), "width": float(rng.uniform(1.5, 5)), "free_flow_speed": float(rng.uniform(0.9, 1.6)),
               "k_critical": float(rng.uniform(1.2, 2.5)), "k_jam": float(rng.uniform(4.5, 7)), "gamma": float(rng.choice([0, 0.005, 0.01, 0.02])),
               "speed_noise_std": float(rng.choice([0, 0.03, 0.05])), "fd_type": fd, "bi_factor": float(rng.choice([1, 1, 1.2, 1.5])),
               "activity_probability": float(rng.choice([0, 0, 0.1, 0.3]))}
    links = {}
    pairs = [(i, j) for i in range(n) for j in range(i + 1, n) if adj[i, j]]
    for (i, j) in pairs:
        if rng.random() < 0.3:
            links[f"{i}_{j}"] = {"length": float(rng.uniform(35, 150)), "fd_type": str(rng.choice(["yperman", "greenshields", "smulders"]))}
        if rng.random() < 0.1:
            links.setdefault(f"{i}_{j}", {})["controller_type"] = "separator"
    if short_links:
        srng = np.random.default_rng(seed + 7919)          # its own stream: the ordinary cases stay what they were
        for (i, j) in pairs:
            if srng.random() < 0.35:
                links.setdefault(f"{i}_{j}", {})["length"] = float(srng.uniform(0.6, 0.55 * dt))   # tt0 = length / v_f of a few seconds
    k = int(rng.integers(1, 4))
    nodes = rng.permutation(n)
    origins = [int(x) for x in nodes[:k]]
    with_od = rng.random() < 0.7
    dests = [int(x) for x in nodes[k:k + int(rng.integers(1, 4))]] if with_od else []
    params = {"unit_time": dt, "simulation_steps": T, "assign_flows_type": "classic", "seed": int(seed),
              "path_finder": {"k_paths": int(rng.integers(1, 5)), "temp": float(rng.uniform(1, 10)), "alpha": float(rng.uniform(0.5, 5)),
                              "beta": float(rng.uniform(0.2, 2)), "omega": float(rng.uniform(0.2, 5))},
              "default_link": default, "links": links,
              "demand": {f"origin_{o}": {"pattern": str(rng.choice(["gaussian_peaks", "constant", "sudden_demand"])),
                                         "peak_lambda": float(rng.uniform(10, 60)), "base_lambda": float(rng.uniform(3, 30))} for o in origins}}
    if with_od and rng.random() < 0.5:
        deg = adj.sum(axis=0)
        cand = [int(x) for x in np.flatnonzero(deg >= 3) if x not in origins and x not in dests]
        if cand:
            params["controllers"] = {"enabled": True, "nodes": cand[:2]}
    return adj, params, origins, dests
