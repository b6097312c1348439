"""Scenario configuration loading (host side of the drop-in boundary).

Mirrors the interface of the reference's ``load_config`` (/root/reference/src/utils/config.py:5-51):
a YAML scenario file becomes ``{'params', 'origin_nodes', 'destination_nodes'[, 'adjacency_matrix'][, 'od_flows']}``
where ``params`` is the dict the ``Network`` constructor consumes.  In addition a neutral JSON form of the
same content (``scenario.json``, written by tools/import_scenarios.py) is accepted so that scenario fixtures
can be shipped without a YAML parser on the machine.
"""
import json

import numpy as np


def _assemble(raw: dict) -> dict:
    sim = raw["simulation"]
    params = {
        "simulation_steps": sim["simulation_steps"],
        "unit_time": sim["unit_time"],
        "assign_flows_type": sim.get("assign_flows_type", "classic"),
        "seed": sim.get("seed", None),
        "path_finder": sim.get("path_finder", {}),
        "default_link": raw["default_link"],
        "links": raw.get("links", {}),
        "demand": raw.get("demand", {}),
        "controllers": raw.get("controllers", {}),
    }
    net = raw["network"]
    cfg = {
        "params": params,
        "origin_nodes": net["origin_nodes"],
        "destination_nodes": net.get("destination_nodes", []),
    }
    if "adjacency_matrix" in net:
        cfg["adjacency_matrix"] = np.array(net["adjacency_matrix"])
    if "od_flows" in raw:
        cfg["od_flows"] = {tuple(int(x) for x in key.split("_")): flow for key, flow in raw["od_flows"].items()}
    return cfg


def load_config(config_path: str) -> dict:
    """Read a scenario file (``*.yaml`` / ``*.yml`` or ``*.json``) into the config dict."""
    if str(config_path).endswith(".json"):
        with open(config_path, "r") as f:
            raw = json.load(f)
    else:
        import yaml

        with open(config_path, "r") as f:
            raw = yaml.safe_load(f)
    return _assemble(raw)


def validate_config(config: dict) -> None:
    """Raise ``ValueError`` when a required section/field of a raw scenario dict is missing
    (same required set as /root/reference/src/utils/config.py:53-77)."""
    required = {
        "network": ["origin_nodes"],
        "simulation": ["simulation_steps", "unit_time"],
        "default_link": ["length", "width", "free_flow_speed", "k_critical", "k_jam"],
    }
    for section, fields in required.items():
        if section not in config:
            raise ValueError(f"Missing required section: {section}")
        for field in fields:
            if field not in config[section]:
                raise ValueError(f"Missing required field: {field} in section {section}")
