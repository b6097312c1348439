#!/usr/bin/env python3
"""bench.py -- link-updates/s of the network_loading hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one network_loading(t) over the whole replica batch of one GPU.  Workload at N = 1: the Melbourne network
(341 nodes, 938 directed links, T = 500) x 1024 concurrent replicas, every replica with its own RNG key and its own
Poisson origin demand, histories resident in HBM in full-record mode.  N > 1: the replica ensemble is sharded, 1024
replicas per GPU (weak scaling), no collective on the step path; torch.distributed (RCCL) is used only for the barriers
and the max-over-ranks reduction of the wall time.

value = links x replicas(all ranks) x K / wall-seconds (max over ranks) of the timed region; inputs are resident in
HBM before the region starts.  The line also carries
  roofline     dominant kernel (node_kernel): algorithmic bytes per launch / HIP-event duration vs 8 TB/s
  cpu_baseline the C restatement under oracle/ timed on this host's cores on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: 8.0 TB/s spec
BYTES_PER_LINK_UPDATE = 212      # 128 B read + 84 B written per link-update, full-record mode
NODE_KERNEL_BYTES = 164          # of which sending/receiving/solve/cumulative update (node_kernel): 116 B read + 48 B written
LINK_KERNEL_BYTES = 48           # density / speed / travel-time update (link_kernel): 12 B read + 36 B written


def replica_demand(T, key, base=5.0, peak=10.0):
    """Per-replica origin demand: Poisson around the reference's gaussian-peaks profile (od_manager.py:145-155 shape)."""
    t = np.arange(T)
    lam = base + peak * np.exp(-(t - T / 4) ** 2 / (2 * (T / 20) ** 2)) + peak * np.exp(-(t - 3 * T / 4) ** 2 / (2 * (T / 20) ** 2))
    return np.random.default_rng(1000 + key).poisson(lam).astype(np.float64)


def cpu_baseline(model, net, origin_nodes, threads, seconds_target=12.0):
    """Oracle (C restatement) on the host cores: `threads` replicas in parallel, full episodes, until ~seconds_target."""
    import oracle_driver as od

    T = int(model["T"])
    oracles = [od.Oracle(model, seed=0, replica=i) for i in range(threads)]
    done, t0 = 0, time.perf_counter()
    rounds = 0
    while True:
        for i, o in enumerate(oracles):
            key = rounds * threads + i
            o.reset(seed=0, replica=key)
            for nid in origin_nodes:
                o.set_demand(net.nodes[nid].index, replica_demand(T, key))
        od.run_many(oracles, 1, T)
        done += threads
        rounds += 1
        el = time.perf_counter() - t0
        if el >= seconds_target or rounds >= 64:
            break
    for o in oracles:
        o.close()
    lu = done * (T - 1) * int(model["n_links"])
    return {"value": lu / el, "unit": "link-updates/s", "cores": threads, "kind": "port",
            "sample": f"{done} replicas x {T - 1} steps of melbourne on {threads} host threads (oracle/pedn_oracle.c, {el:.1f} s)"}


def measured_traffic(kernel="node_kernel", network="melbourne", replicas=1024):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN_pmc.json, written by
    tools/summarize_profiles.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command).  bench.py cannot
    collect PMC counters on itself; the number is only reported for the workload it was measured on."""
    import glob
    if network != "melbourne" or replicas != 1024:
        return None, None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    k = d.get("kernels", {}).get(kernel)
    return (k["hbm_bytes_per_launch"], os.path.relpath(files[-1], ROOT)) if k else (None, None)


def bench_rl(args):
    """BASELINE config #5: env-steps/s of the batched RL step (action clipping + network_loading + observations + rewards)
    on 45_intersections x 2048 envs, obs option3, action_gap 1, uniform random actions resident in HBM."""
    import torch

    from pednstream_amd.rl_env import VecPedNetEnv

    B = args.replicas
    env = VecPedNetEnv(args.network, n_envs=B, obs_mode="option3", action_gap=1, seed=0, data_dir=os.path.join(ROOT, "data"))
    e = env.network.engine()
    T = env.simulation_steps
    K = min(args.steps, T - 1 - args.warmup)
    gen = torch.Generator(device="cuda").manual_seed(0)
    hi = torch.as_tensor(env.action_high, device="cuda", dtype=torch.float64)
    acts = torch.rand((args.warmup + K, B, env.n_actions), generator=gen, device="cuda", dtype=torch.float64) * hi
    torch.cuda.synchronize()
    row = B * env.n_actions * 8
    t = 1
    for k in range(args.warmup):
        e.rl_step_device(acts.data_ptr() + k * row, t)
        t += 1
    e.synchronize()
    e.timer_begin()
    t0 = time.perf_counter()
    for k in range(args.warmup, args.warmup + K):
        e.rl_step_device(acts.data_ptr() + k * row, t)
        t += 1
    dev_ms = e.timer_end()
    wall = time.perf_counter() - t0
    rc, _ = e.error_flags()
    assert rc == 0
    out = {"metric": "env-steps/sec (replicas x steps/sec, incl. action apply, observations, rewards)", "value": B * K / wall,
           "unit": "env-steps/s", "n_gpus": 1, "steps": K, "warmup": args.warmup, "ms_per_step": wall / K * 1e3,
           "device_ms_per_step": dev_ms / K, "higher_is_better": True, "data": "synthetic", "dtype": "f64+f32",
           "link_updates_per_s": e.n_links * B * K / wall,
           "config": {"workload": f"{args.network} x {B} envs, obs option3 ({env.n_obs} floats), {env.n_actions} action dims, "
                                  f"agents {env.possible_agents}, actions resident in HBM (torch), obs/rewards left on device"}}
    print(json.dumps(out), flush=True)
    env.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--network", default="melbourne")
    ap.add_argument("--replicas", type=int, default=1024, help="replicas per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                    "multi-rank path on one GPU)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--rng-mode", default="philox", choices=["philox", "meanfield"], help="diagnostic: meanfield removes the RNG work")
    ap.add_argument("--rl", action="store_true", help="config #5 instead: batched RL env step, env-steps/s")
    args = ap.parse_args()
    if args.rl:
        return bench_rl(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.share_device:
        local_rank = 0
    dist = None
    if "RANK" in os.environ:          # launched by torch.distributed.run, also for N = 1
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    from pednstream_amd import NetworkEnvGenerator
    from pednstream_amd.flatten import flatten_network

    from pednstream_amd.ensemble import shard

    R = args.replicas                                          # weak scaling: R replicas on every GPU
    offset, count = shard(R * world, world, rank)              # contiguous block of global replica ids
    assert count == R
    gen = NetworkEnvGenerator(os.path.join(ROOT, "data"))
    net = gen.create_network(args.network, verbose=False, n_replicas=R, replica_offset=offset, rng_seed=0,
                             rng_mode=args.rng_mode, device=local_rank)
    T = net.simulation_steps
    e = net.engine()
    origins = list(net.origin_nodes)
    for k, nid in enumerate(origins):          # one upload per origin: [R, T] rows keyed by the global replica id
        e.set_demand_matrix(net.nodes[nid].index, np.stack([replica_demand(T, offset + r) for r in range(R)]))
    e.synchronize()
    L = e.n_links

    def barrier():
        e.synchronize()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()
        e.synchronize()

    # the simulation clock runs 1..T-1; an episode that reaches T is reset and continues (reset is inside the timed region)
    state = {"t": 1}

    def advance(n):
        left = n
        while left > 0:
            if state["t"] >= T:
                e.reset()
                state["t"] = 1
            k = min(left, T - state["t"])
            e.run(state["t"], state["t"] + k)
            state["t"] += k
            left -= k

    advance(args.warmup)
    barrier()
    e.timer_begin()
    t0 = time.perf_counter()
    advance(args.steps)
    dev_ms = e.timer_end()          # HIP events on the engine's stream
    barrier()
    wall = time.perf_counter() - t0
    if dist is not None:
        import torch
        w = torch.tensor([wall], device="cuda" if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
    rc, _ = e.error_flags()
    if rc != 0:
        raise SystemExit(f"model error flags set: {rc}")

    # per-kernel durations (HIP events around each launch, same stream), continuing the same simulation
    prof = []
    for _ in range(min(40, max(8, args.steps // 8))):
        if state["t"] >= T:
            e.reset()
            state["t"] = 1
        prof.append(e.profile_step(state["t"]))
        state["t"] += 1
    prof = np.array(prof)
    tf_ms, node_ms, link_ms = prof.mean(axis=0)

    if rank == 0:
        total_lu = L * R * world * args.steps
        value = total_lu / wall
        node_bytes = NODE_KERNEL_BYTES * L * R
        achieved = node_bytes / (node_ms * 1e-3) / 1e9
        traffic, traffic_src = measured_traffic("node_kernel", args.network, R)
        out = {
            "metric": "link-updates/sec (links x replicas x steps/sec)", "value": value, "unit": "link-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64+f32", "data": "synthetic",
            "config": {"workload": f"{args.network} network ({L} links, {len(net.nodes)} nodes, T={T}) x {R} replicas per GPU, "
                                   f"full-record mode, per-replica Poisson demand and Philox keys",
                       "replicas_per_gpu": R, "links": L, "parallelism": f"replica-sharded x{world}, no step-path collective"},
            "device_ms_per_step": dev_ms / args.steps,
            "roofline": {"bound": "hbm", "kernel": "node_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": node_bytes, "avg_launch_ms": float(node_ms),
                         "other_kernels_ms": {"link_kernel": float(link_ms), "turn_prob_kernel": float(tf_ms)},
                         "whole_step_GBps": BYTES_PER_LINK_UPDATE * L * R / ((node_ms + link_ms + tf_ms) * 1e-3) / 1e9},
        }
        if not args.no_cpu_baseline and world == 1:
            threads = min(os.cpu_count() or 1, 16)
            out["cpu_baseline"] = cpu_baseline(flatten_network(net), net, origins, threads)
        print(json.dumps(out), flush=True)
    net.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
