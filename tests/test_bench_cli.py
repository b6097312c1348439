"""bench.py without a GPU: the command line the driver uses parses, the synthetic demand is a function of the replica id only
(it keys results to the GLOBAL replica id, whatever the rank count), and a WORLD_SIZE that disagrees with --gpus is refused
before anything is imported."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def test_help_lists_the_contract_flags():
    out = subprocess.run([sys.executable, BENCH, "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--replicas", "--total-replicas", "--network", "--rl", "--randomize", "--history"):
        assert flag in out.stdout, flag


def test_world_size_mismatch_is_refused_early():
    env = dict(os.environ, WORLD_SIZE="4")
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr


def test_replica_demand_is_keyed_by_the_global_replica_id():
    sys.path.insert(0, ROOT)
    import bench

    a, b = bench.replica_demand(500, 7), bench.replica_demand(500, 7)
    assert np.array_equal(a, b) and a.dtype == np.float64 and len(a) == 500
    assert not np.array_equal(a, bench.replica_demand(500, 8))
    assert bench.BYTES_PER_LINK_UPDATE == bench.NODE_KERNEL_BYTES + bench.LINK_KERNEL_BYTES == 212


def _full_object():
    """A full bench object of the shape main() builds (every block, long prose strings where the real one has them)."""
    prose = "x" * 600
    rf = {"bound": "hbm", "kernel": prose, "kernel_name": "node_kernel<LU>", "achieved": 5441.1859, "peak": 8000.0, "unit": "GB/s", "frac": 0.68014821,
          "basis_kind": "moved", "basis": prose, "traffic": 80879083.2, "traffic_source": prose, "frac_algorithmic": 0.8562, "frac_counter": 0.6801,
          "frac_executed": 0.5931, "traffic_bytes_per_link_update": 168.4, "algorithmic_bytes_per_link_update": 212,
          "algorithmic_bytes_per_launch": 101814272, "fits_infinity_cache": True, "concurrent_chains": 2, "avg_launch_ms": 0.02592,
          "whole_step_frac": 0.8484, "whole_step_frac_counter": 0.674, "note": prose, "per_launch": {"note": prose},
          "launch_plan": {"chains": 2, "link_update_by_next_node_kernel": True, "probe": prose}, "one_chain": {"avg_launch_ms": 0.0322, "frac": 0.79, "note": prose}}
    variant = {"value": 1.9e10, "unit": "link-updates/s", "ms_per_step": 0.0435, "device_ms_per_step": 0.0434, "steps": 300, "warmup": 100,
               "config": {"workload": prose}, "roofline": dict(rf), "cpu_baseline": {"sample": prose}}
    rl = {"value": 8.1e7, "ms_per_step": 0.0252, "whole_step_frac": 0.362, "config": {"workload": prose}}
    return {"metric": "link-updates/sec (links x replicas x steps/sec)", "value": 31951388757.4, "unit": "link-updates/s", "n_gpus": 1, "ranks_seen": 1,
            "steps": 20, "warmup": 5, "ms_per_step": 0.0300616, "device_ms_per_step": 0.02999, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64+f32", "data": "synthetic",
            "config": {"workload": prose, "network": "melbourne", "history": "full-record", "replicas_per_gpu": 1024, "replicas_total": 1024, "links": 938,
                       "parallelism": prose},
            "roofline": rf,
            "cpu_baseline": {"value": 8.4e7, "unit": "link-updates/s", "cores": 256, "kind": "port", "sample": prose,
                             "sample_short": "1280 replicas x 499 steps of melbourne, 256 threads, 7.1 s; 1 thread: 54 replicas, 7.0 s", "one_thread": 3.5e6,
                             "all_cores": 8.4e7, "cpu_model": "AMD EPYC 9575F 64-Core Processor", "executed_paths": {"a": 1},
                             "reference_python_equiv": {"label": prose, "one_core": 27920.1, "all_cores": 6.6e5}},
            "extra": {"headline_long_window": dict(variant), "config3_delft_x1024": dict(variant), "hbm_proper_melbourne_x4096": dict(variant),
                      "melbourne_heavy_x1024": dict(variant), "nine_x256": dict(variant), "broken": {"error": prose},
                      "config5_rl_45int_x2048": {"plain": dict(rl), "randomized": dict(rl),
                                                 "by_n_envs_recent_history": {str(n): {"value": 1e8} for n in (1024, 2048, 4096, 8192)},
                                                 "end_to_end_random_policy": {"eager": {"value": 3.2e7}, "graph_replay": {"value": 6e7}},
                                                 "end_to_end_mlp_policy": {"eager": {"value": 1.3e7}, "graph_replay": {"value": 5e7}},
                                                 "sustained": {f"{m}_{h}": {"value": 7e7, "wall_s": 1.0} for h in ("full", "recent")
                                                               for m in ("plain", "vectorised", "reference")}}}}


def test_compact_line_is_small_and_carries_the_contract_fields():
    """the driver keeps an 8 KB tail of stdout and parses its last line -- the line must stay under 4 KB whatever the
    run measured, and carry metric / value / roofline.frac / cpu_baseline.value as numbers."""
    import json

    sys.path.insert(0, ROOT)
    import bench

    line = json.dumps(bench.compact_line(_full_object()), separators=(",", ":"))
    assert len(line) < 4096, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["config"]["workload"].startswith("melbourne x 1024") and "model" not in d["config"]
    rf = d["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "basis_kind", "traffic", "frac_algorithmic", "frac_counter", "frac_executed",
              "avg_launch_ms", "launch_plan"):
        assert k in rf, k
    assert rf["kernel"] == "node_kernel<LU>" and 0 < rf["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] == 256 and cb["kind"] == "port" and len(cb["sample"]) < 120
    assert d["extra"]["config5_rl_45int_x2048"]["end_to_end_mlp_policy"]["graph_replay"] == 5e7
    assert all(len(json.dumps(v)) < 900 for v in d["extra"].values())
    # no prose anywhere
    assert "xxxx" not in line.replace(d["extra"]["broken"]["error"], "")
