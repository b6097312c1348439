"""Origin-destination weights and origin demand generation (host side, set-up time only).

Public surface mirrored from /root/reference/src/LTM/od_manager.py: ``ODManager`` (:14-54: ``init_od_flows``,
``get_od_flow``, attribute ``od_flows``) and ``DemandGenerator`` (:57-155: pattern registry, ``generate_custom`` and the
three built-in patterns).  The built-in patterns consume numpy's *global* RNG in the reference's order -- including the
reseed with ``simulation.seed`` right before the Poisson draw (:153-154) -- so a seeded scenario yields the reference's
demand arrays under the same numpy.  The arrays are uploaded to the device once; the per-step lookup of the OD weights
(``get_od_flow``) happens on the device / in the host-side tabulation of P(od | up).
"""
import logging
from dataclasses import dataclass

import numpy as np

_LOG = logging.getLogger(__name__)


@dataclass
class DemandConfig:
    """Per-origin demand settings with the reference's fall-back values."""

    peak_lambda: float = 10.0
    base_lambda: float = 5.0
    seed: int = 42
    pattern: str = "gaussian_peaks"


class ODManager:
    """``od_flows[(origin, destination)]`` = relative destination weight per time index (length T+1)."""

    def __init__(self, simulation_steps: int, logger: logging.Logger = None):
        self.logger = logger or _LOG
        self.simulation_steps = simulation_steps
        self.od_flows = {}
        self._zeros = np.zeros(simulation_steps + 1)

    def init_od_flows(self, origin_nodes, destination_nodes, od_flows: dict = None):
        n = self.simulation_steps + 1
        if not od_flows:
            self.logger.info("No OD flows provided, initializing with ones")
            self.od_flows.update({(o, d): np.ones(n) for o in origin_nodes for d in destination_nodes if o != d})
            return
        for (o, d), weight in od_flows.items():
            if isinstance(weight, (int, float)):
                self.od_flows[(o, d)] = np.full(n, weight)
            elif len(weight) == n:
                self.od_flows[(o, d)] = np.array(weight)
            else:
                raise ValueError(f"Flow array length for OD pair ({o},{d}) must match simulation_steps")

    def get_od_flow(self, origin: int, destination: int, time_step: int) -> float:
        return self.od_flows.get((origin, destination), self._zeros)[time_step]

    def as_matrix(self) -> np.ndarray:
        """[n_od, T+1] binary64 in dict order -- the layout uploaded to the device."""
        rows = [np.asarray(w, dtype=np.float64) for w in self.od_flows.values()]
        return np.stack(rows) if rows else np.zeros((0, self.simulation_steps + 1))


class DemandGenerator:
    """Named demand patterns for origin nodes; custom callables ``f(origin_id, params=...)`` can be registered."""

    def __init__(self, simulation_steps: int, params: dict, logger: logging.Logger = None):
        self.logger, self.simulation_steps, self.params = logger, simulation_steps, params
        self.time = np.arange(simulation_steps)
        self.seed = params.get("seed", None)
        self.demand_patterns = {"gaussian_peaks": self.generate_gaussian_peaks, "constant": self.generate_constant,
                                "sudden_demand": self.generate_sudden_demand}

    def register_pattern(self, pattern_name: str, pattern_func):
        if not callable(pattern_func):
            raise ValueError("pattern_func must be callable")
        self.demand_patterns[pattern_name] = pattern_func

    def generate_custom(self, origin_id: int, pattern: str) -> np.ndarray:
        try:
            fn = self.demand_patterns[pattern]
        except KeyError:
            raise ValueError(f"Unknown demand pattern: {pattern}. Available patterns: {list(self.demand_patterns.keys())}") from None
        return fn(origin_id, params=self.params)

    # ---- built-in patterns ---------------------------------------------------------------------------------------
    def _get_demand_config(self, origin_id: int) -> DemandConfig:
        entry = self.params.get("demand", {}).get(f"origin_{origin_id}") if "demand" in self.params else None
        if entry is None:
            if self.logger:
                self.logger.info(f"No demand configuration found for origin {origin_id}, using defaults")
            return DemandConfig()
        return DemandConfig(entry.get("peak_lambda", 10.0), entry.get("base_lambda", 5.0), self.seed,
                            entry.get("pattern", "gaussian_peaks"))

    def _two_peak_poisson(self, cfg: DemandConfig) -> np.ndarray:
        """Poisson counts around base + two Gaussian bumps at T/4 and 3T/4 of width T/20 (od_manager.py:145-155)."""
        T = self.simulation_steps
        bumps = [cfg.peak_lambda * np.exp(-(self.time - centre) ** 2 / (2 * (T / 20) ** 2)) for centre in (T / 4, 3 * T / 4)]
        rate = cfg.base_lambda + bumps[0] + bumps[1]
        if self.seed is not None:
            np.random.seed(self.seed)
        return np.random.poisson(lam=rate)

    def generate_gaussian_peaks(self, origin_id: int, params=None) -> np.ndarray:
        return self._two_peak_poisson(self._get_demand_config(origin_id))

    def generate_constant(self, origin_id: int, params=None) -> np.ndarray:
        return np.full(self.simulation_steps + 1, self._get_demand_config(origin_id).base_lambda)

    def generate_sudden_demand(self, origin_id: int, params=None) -> np.ndarray:
        counts = self._two_peak_poisson(self._get_demand_config(origin_id))
        length = np.random.randint(10, 20)                                        # three global-RNG draws, in this order
        first = np.random.randint(0, max(1, self.simulation_steps - length))
        counts[first:first + length] += np.random.randint(20, 50)
        return counts
