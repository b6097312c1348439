#!/usr/bin/env python3
"""CPU calibration of BASELINE.md section 4 (runs only in the build container, which has /root/reference).

Times, on the SAME host cores and the same scenario,
  (i)  the reference's own Python path -- `for t in range(1, T): network.network_loading(t)`
       (/root/reference/src/LTM/network.py:266-287), one process, one core, native numpy RNG, construction excluded;
  (ii) the build's C restatement (oracle/pedn_oracle.c), 1 thread and all cores (one replica per thread),
and writes the restatement / reference speed ratio per network to profiles/cpu_calibration.json (tracked).  bench.py,
which runs where the reference cannot travel, times only (ii) on the GPU box's host cores and divides by this ratio to
report a *derived* reference-Python equivalent.

    python tools/calibrate_cpu.py [melbourne delft ...]
"""
import json
import os
import platform
import sys
import time

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
os.environ.setdefault("MPLBACKEND", "Agg")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or "unknown"


def time_reference(name, max_steps=None):
    import ref_harness as rh

    ref = rh.load_reference()
    np.random.seed(1)
    net = ref["env"].NetworkEnvGenerator().create_network(name)
    T = net.params["simulation_steps"]
    last = T if max_steps is None else min(T, max_steps + 1)
    t0 = time.perf_counter()
    for t in range(1, last):
        net.network_loading(t)
    el = time.perf_counter() - t0
    return len(net.links) * (last - 1) / el, last - 1, el


def time_oracle(name, threads, seconds=4.0, max_steps=None):
    import oracle_driver as od
    from pednstream_amd import NetworkEnvGenerator
    from pednstream_amd.flatten import flatten_network

    np.random.seed(1)
    net = NetworkEnvGenerator(os.path.join(ROOT, "data")).create_network(name, verbose=False)
    model = flatten_network(net)
    T, L = int(model["T"]), int(model["n_links"])
    last = T if max_steps is None else min(T, max_steps + 1)      # the same step range as the reference run
    oracles = [od.Oracle(model, seed=0, replica=i) for i in range(threads)]
    done, rounds, t0 = 0, 0, time.perf_counter()
    while True:
        for i, o in enumerate(oracles):
            o.reset(seed=0, replica=rounds * threads + i)
        od.run_many(oracles, 1, last)
        done += threads
        rounds += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    for o in oracles:
        o.close()
    return L * done * (last - 1) / el, done, el


def main():
    names = sys.argv[1:] or ["melbourne", "delft"]
    cores = os.cpu_count() or 1
    out = {"host": {"cpu_model": cpu_model(), "cores_total": cores, "python": platform.python_version(), "numpy": np.__version__},
           "method": "reference: /root/reference Network.network_loading loop, 1 process, native RNG; port: oracle/pedn_oracle.c "
                     "(gcc -O2 -fopenmp), one replica per thread; same scenario yaml, default demand",
           "networks": {}}
    for name in names:
        ms = None if name != "delft" else 150      # delft takes 99 s per episode in the reference: first 150 steps, both sides
        ref_lu, ref_steps, ref_s = time_reference(name, max_steps=ms)
        o1, n1, s1 = time_oracle(name, 1, max_steps=ms)
        oa, na, sa = time_oracle(name, cores, max_steps=ms)
        out["networks"][name] = {
            "reference_python_link_updates_per_s_1core": ref_lu, "reference_steps_timed": ref_steps, "reference_seconds": ref_s,
            "port_link_updates_per_s_1thread": o1, "port_link_updates_per_s_allcores": oa, "port_threads_allcores": cores,
            "port_over_reference_1core": o1 / ref_lu}
        print(f"{name}: reference {ref_lu:.3e} lu/s (1 core, {ref_steps} steps in {ref_s:.1f} s); port {o1:.3e} (1 thread), "
              f"{oa:.3e} ({cores} threads); ratio {o1 / ref_lu:.1f}x", flush=True)
    path = os.path.join(ROOT, "profiles", "cpu_calibration.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
