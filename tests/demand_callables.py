"""Custom demand callables for the boundary paths `Network(..., demand_pattern=[fn])` (reference: src/LTM/network.py:88-93,
dispatch by `fn.__name__` through DemandGenerator.generate_custom, src/LTM/od_manager.py:125-143) and
`NetworkEnvGenerator.create_network(name, [fn])` (src/utils/env_loader.py:81-85,147-156).  TEST INFRASTRUCTURE: the golden
generator (oracle/gen_golden.py) hands these to the REAL reference, the tests hand them to this repository's host code; the
fixtures hold what came out.  They are this repository's own callables in the two shapes the reference's examples use
(examples/spike.py: a pattern that draws from the global numpy RNG and reads the yaml-style demand section;
examples/Melbourne.py: a closure over a table of counts per minute, expanded to simulation steps)."""
import numpy as np


def plateau_pattern(origin_id, params):
    """Poisson counts (GLOBAL numpy RNG, like the built-in patterns) around one early bump, a flat stretch of fixed demand, and
    nothing after three quarters of the horizon.  Length = simulation_steps (one short of the history length), integer dtype."""
    cfg = params["demand"][f"origin_{origin_id}"]
    T = params["simulation_steps"]
    t = np.arange(T)
    lam = cfg["base_lambda"] + cfg["peak_lambda"] * np.exp(-(t - T / 5) ** 2 / (2 * (T / 16) ** 2))
    demand = np.random.poisson(lam=lam)
    demand[T // 2:T // 2 + 25] = 24
    demand[3 * T // 4:] = 0
    return demand


def make_table_demand(per_minute, steps_per_minute=6):
    """A callable bound to a table {origin node: counts per minute}: every minute's count spread evenly over its steps, rounded
    up.  The result is as long as the table makes it -- not the horizon -- and float64."""
    def node_demand_from_table(origin_node, params=None, _table=per_minute, _k=steps_per_minute):
        counts = np.asarray(_table[origin_node], dtype=np.float64)
        return np.ceil(np.repeat(counts / _k, _k))
    return node_demand_from_table


# the table of the melbourne_callable golden: 85 minutes (510 steps > T + 1 = 501) for the yaml's origin
MELBOURNE_TABLE = {289: [int(40 + 35 * np.sin(i / 7.0) ** 2 + (i * 37) % 11) for i in range(85)]}

SPIKE_ADJ = [[0, 1, 1, 1, 0, 0],
             [1, 0, 1, 1, 0, 0],
             [1, 1, 0, 1, 0, 0],
             [1, 1, 1, 0, 1, 0],
             [0, 0, 0, 1, 0, 1],
             [0, 0, 0, 0, 1, 0]]


def spike_params():
    return {"unit_time": 10, "simulation_steps": 400, "assign_flows_type": "classic", "custom_pattern": "plateau_pattern",
            "default_link": {"length": 100, "width": 2, "free_flow_speed": 1.1, "k_critical": 2, "k_jam": 6, "speed_noise_std": 0,
                             "fd_type": "yperman", "bi_factor": 1, "controller_type": "gate"},
            "demand": {"origin_4": {"pattern": "plateau_pattern", "peak_lambda": 20, "base_lambda": 10}}}
