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
