"""Convert the reference's scenario inputs (data only: YAML parameters, adjacency, edge lengths, node positions)
into the neutral fixture form shipped in this repository: data/<name>/scenario.json + data/<name>/topology.npz.

Run once in the build container:  python tools/import_scenarios.py [/root/reference/data] [data]
The converted files hold scenario DATA (numbers and names); no reference source code is copied.
"""
import json
import os
import pickle
import sys

import numpy as np
import yaml


def convert(src_dir, dst_dir):
    os.makedirs(dst_dir, exist_ok=True)
    with open(os.path.join(src_dir, "sim_params.yaml")) as f:
        raw = yaml.safe_load(f)
    with open(os.path.join(dst_dir, "scenario.json"), "w") as f:
        json.dump(raw, f, indent=1, sort_keys=False)
    topo = {}
    adj_path = os.path.join(src_dir, "adj_matrix.npy")
    if os.path.exists(adj_path):
        adj = np.load(adj_path)
        rows, cols = np.nonzero(adj == 1)
        topo.update(n_nodes=np.int64(adj.shape[0]), adj_rows=rows.astype(np.int32), adj_cols=cols.astype(np.int32))
    ed_path = os.path.join(src_dir, "edge_distances.pkl")
    if os.path.exists(ed_path):
        with open(ed_path, "rb") as f:
            ed = pickle.load(f)
        # dict order is kept: it decides which direction's parameters win in create_network
        topo.update(edge_uv=np.array([[u, v] for (u, v) in ed.keys()], dtype=np.int32),
                    edge_dist=np.array([float(d) for d in ed.values()], dtype=np.float64))
    pos_path = os.path.join(src_dir, "node_positions.json")
    if os.path.exists(pos_path):
        with open(pos_path) as f:
            pos = json.load(f)
        topo.update(pos_ids=np.array([int(k) for k in pos.keys()], dtype=np.int64),
                    pos_xy=np.array([[float(v[0]), float(v[1])] for v in pos.values()], dtype=np.float64))
    if topo:
        np.savez_compressed(os.path.join(dst_dir, "topology.npz"), **topo)


if __name__ == "__main__":
    src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/data"
    dst = sys.argv[2] if len(sys.argv) > 2 else "data"
    for name in sorted(os.listdir(src)):
        if os.path.exists(os.path.join(src, name, "sim_params.yaml")):
            convert(os.path.join(src, name), os.path.join(dst, name))
            print("converted", name)
