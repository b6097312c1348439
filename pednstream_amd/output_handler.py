"""Simulation dump in the reference's on-disk format.

Mirrors /root/reference/handlers/output_handler.py: ``OutputHandler(base_dir, simulation_dir)``,
``save_network_state(network)`` -> ``link_data.json`` / ``node_data.json`` / ``network_params.json`` (:27-93),
``save_time_series`` -> ``time_series.csv`` (:95-118), ``load_simulation`` (:126-148), so that the reference's
``NetworkVisualizer(simulation_dir=...)`` keeps working on runs of the MI355X engine.  Histories are pulled from the
device one field at a time (one strided gather per field) instead of one link attribute at a time.
"""
import json
from datetime import datetime
from pathlib import Path

import numpy as np

_LINK_ARRAYS = ("density", "link_flow", "speed", "travel_time", "inflow", "outflow", "num_pedestrians",
                "cumulative_inflow", "cumulative_outflow", "sending_flow", "receiving_flow")


class OutputHandler:
    def __init__(self, base_dir="outputs", simulation_dir=None):
        self.base_dir = Path(base_dir)
        self.timestamp = datetime.now().strftime("%Y%m%d_%H%M%S")
        self.simulation_dir = self.base_dir / (simulation_dir if simulation_dir is not None else f"sim_{self.timestamp}")
        self.simulation_dir.mkdir(parents=True, exist_ok=True)

    @staticmethod
    def _field_columns(network, name, replica):
        """[n_links, T+1] history of one field for one replica."""
        if hasattr(network, "read_field"):
            with network.replica(replica):
                eng = network._flush()
            from .network import LINK_FIELDS

            block = eng.read_block(LINK_FIELDS[name][0], 0, network.simulation_steps + 1, rep0=replica, rep1=replica + 1)
            return block[:, :network.n_links, 0].T
        return np.stack([np.asarray(getattr(l, name)) for l in network.links.values()])

    def build_network_state(self, network, replica=0):
        """The three dictionaries ``save_network_state`` writes."""
        cols = {name: self._field_columns(network, name, replica) for name in _LINK_ARRAYS}
        gate = self._field_columns(network, "back_gate_width_data", replica)
        link_data = {}
        gaters = getattr(network, "controller_gaters", set())
        for i, ((u, v), link) in enumerate(network.links.items()):
            entry = {name: cols[name][i].tolist() for name in _LINK_ARRAYS}
            entry["parameters"] = {"length": link.length, "width": link.width, "free_flow_speed": link.free_flow_speed,
                                   "k_critical": link.k_critical, "k_jam": link.k_jam}
            if u in gaters:
                entry["back_gate_width"] = gate[i].tolist()
            if getattr(link, "is_separator", False):
                entry["is_separator"] = True
                entry["separator_width"] = np.asarray(link.separator_width_data).tolist()
            link_data[f"{u}-{v}"] = entry
        node_data = {}
        for node in network.nodes.values():
            d = node.demand
            node_data[node.node_id] = {"demand": np.asarray(d).tolist() if d is not None else [],
                                       "incoming_links": [l.link_id for l in node.incoming_links],
                                       "outgoing_links": [l.link_id for l in node.outgoing_links]}
        pf = getattr(network, "path_finder", None)
        network_params = {"simulation_steps": network.simulation_steps, "unit_time": network.unit_time,
                          "destination_nodes": network.destination_nodes, "origin_nodes": network.origin_nodes,
                          "od_paths": ({f"{k[0]}-{k[1]}": v for k, v in pf.od_paths.items()} if pf is not None else {})}
        return link_data, node_data, network_params

    def save_network_state(self, network, replica=0):
        link_data, node_data, network_params = self.build_network_state(network, replica)
        self._save_json(link_data, "link_data.json")
        self._save_json(node_data, "node_data.json")
        self._save_json(network_params, "network_params.json")

    def save_time_series(self, network, replica=0):
        import pandas as pd

        names = ("density", "speed", "inflow", "outflow", "num_pedestrians", "cumulative_inflow", "cumulative_outflow")
        cols = {n: self._field_columns(network, n, replica) for n in names}
        rows = []
        for i, (u, v) in enumerate(network.links.keys()):
            for t in range(network.simulation_steps):
                row = {"time_step": t, "link_id": f"{u}-{v}"}
                row.update({n: cols[n][i, t] for n in names})
                rows.append(row)
        pd.DataFrame(rows).to_csv(self.simulation_dir / "time_series.csv", index=False)

    def _save_json(self, data, filename):
        with open(self.simulation_dir / filename, "w") as f:
            json.dump(data, f, indent=2)

    @staticmethod
    def load_simulation(simulation_dir: str):
        data = {}
        path = Path(simulation_dir)
        for filename in ("link_data.json", "node_data.json", "network_params.json"):
            fp = path / filename
            if fp.exists():
                with open(fp, "r") as f:
                    data[filename.replace(".json", "")] = json.load(f)
        csv = path / "time_series.csv"
        if csv.exists():
            import pandas as pd

            data["time_series"] = pd.read_csv(csv)
        return data
