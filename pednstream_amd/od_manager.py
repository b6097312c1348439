"""Origin-destination weights and origin demand generation (host side, setup time only).

Behavioural mirror of /root/reference/src/LTM/od_manager.py (ODManager :14-54, DemandGenerator :57-155):
same pattern names, same use of numpy's *global* RNG (including the reseed at :153-154) so that a scenario
with ``simulation.seed`` yields the same demand arrays as the reference under the same numpy.
The arrays produced here are uploaded to the device once; the per-step lookup
``get_od_flow`` (:52-54) happens inside the turning-fraction kernel.
"""
import logging
from dataclasses import dataclass
from typing import Callable, Dict

import numpy as np


@dataclass
class DemandConfig:
    peak_lambda: float = 10.0
    base_lambda: float = 5.0
    seed: int = 42
    pattern: str = "gaussian_peaks"


class ODManager:
    """{(o, d): weight array of length T+1}; weights are relative destination preferences."""

    def __init__(self, simulation_steps: int, logger: logging.Logger = None):
        self.logger = logger or logging.getLogger(__name__)
        self.simulation_steps = simulation_steps
        self.od_flows = {}
        self._default_zero_flow = np.zeros(simulation_steps + 1)

    def init_od_flows(self, origin_nodes, destination_nodes, od_flows: dict = None):
        if od_flows:
            for (o, d), flow in od_flows.items():
                if isinstance(flow, (int, float)):
                    self.od_flows[(o, d)] = np.full(self.simulation_steps + 1, flow)
                else:
                    if len(flow) != self.simulation_steps + 1:
                        raise ValueError(f"Flow array length for OD pair ({o},{d}) must match simulation_steps")
                    self.od_flows[(o, d)] = np.array(flow)
            return
        self.logger.info("No OD flows provided, initializing with ones")
        for o in origin_nodes:
            for d in destination_nodes:
                if o != d:
                    self.od_flows[(o, d)] = np.ones(self.simulation_steps + 1)

    def get_od_flow(self, origin: int, destination: int, time_step: int) -> float:
        return self.od_flows.get((origin, destination), self._default_zero_flow)[time_step]

    def as_matrix(self) -> np.ndarray:
        """[n_od, T+1] f64 in dict order -- the layout uploaded to the device."""
        if not self.od_flows:
            return np.zeros((0, self.simulation_steps + 1))
        return np.stack([np.asarray(v, dtype=np.float64) for v in self.od_flows.values()])


class DemandGenerator:
    """Origin demand patterns: gaussian_peaks / constant / sudden_demand + registered callables."""

    def __init__(self, simulation_steps: int, params: dict, logger: logging.Logger = None):
        self.logger = logger
        self.simulation_steps = simulation_steps
        self.params = params
        self.time = np.arange(simulation_steps)
        self.seed = params.get("seed", None)
        self.demand_patterns: Dict[str, Callable] = {
            "gaussian_peaks": self.generate_gaussian_peaks,
            "constant": self.generate_constant,
            "sudden_demand": self.generate_sudden_demand,
        }

    def register_pattern(self, pattern_name: str, pattern_func: Callable):
        if not callable(pattern_func):
            raise ValueError("pattern_func must be callable")
        self.demand_patterns[pattern_name] = pattern_func

    def _get_demand_config(self, origin_id: int) -> DemandConfig:
        try:
            oc = self.params["demand"][f"origin_{origin_id}"]
        except KeyError:
            if self.logger:
                self.logger.info(f"No demand configuration found for origin {origin_id}, using defaults")
            return DemandConfig()
        return DemandConfig(peak_lambda=oc.get("peak_lambda", 10.0), base_lambda=oc.get("base_lambda", 5.0),
                            seed=self.seed, pattern=oc.get("pattern", "gaussian_peaks"))

    def _gaussian_lambda_draw(self, cfg: DemandConfig) -> np.ndarray:
        T = self.simulation_steps
        am = cfg.peak_lambda * np.exp(-(self.time - T / 4) ** 2 / (2 * (T / 20) ** 2))
        pm = cfg.peak_lambda * np.exp(-(self.time - 3 * T / 4) ** 2 / (2 * (T / 20) ** 2))
        lam = cfg.base_lambda + am + pm
        if self.seed is not None:
            np.random.seed(self.seed)
        return np.random.poisson(lam=lam)

    def generate_gaussian_peaks(self, origin_id: int, params=None) -> np.ndarray:
        return self._gaussian_lambda_draw(self._get_demand_config(origin_id))

    def generate_constant(self, origin_id: int, params=None) -> np.ndarray:
        return np.full(self.simulation_steps + 1, self._get_demand_config(origin_id).base_lambda)

    def generate_sudden_demand(self, origin_id: int, params=None) -> np.ndarray:
        demand = self._gaussian_lambda_draw(self._get_demand_config(origin_id))
        period = np.random.randint(10, 20)
        start = np.random.randint(0, max(1, self.simulation_steps - period))
        demand[start:start + period] += np.random.randint(20, 50)
        return demand

    def generate_custom(self, origin_id: int, pattern: str) -> np.ndarray:
        if pattern not in self.demand_patterns:
            raise ValueError(f"Unknown demand pattern: {pattern}. "
                             f"Available patterns: {list(self.demand_patterns.keys())}")
        return self.demand_patterns[pattern](origin_id, params=self.params)
