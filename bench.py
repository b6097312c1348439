#!/usr/bin/env python3
"""bench.py -- link-updates/s of the network_loading hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one network_loading(t) over the whole replica batch of one GPU.  Workload at N = 1: the Melbourne network
(341 nodes, 938 directed links, T = 500) x 1024 concurrent replicas, every replica with its own RNG key and its own
Poisson origin demand, histories resident in HBM in full-record mode.  N > 1: the replica ensemble is sharded over the ranks with
no collective on the step path; torch.distributed (RCCL) is used only for the barriers, the max-over-ranks reduction of the
wall time and the count of ranks that took part.  --replicas R: R replicas on every GPU ("scaling": "weak"; BASELINE config #4's
shape is --replicas 512 on 8 GPUs); --total-replicas M: M replicas in all, M / N per GPU ("scaling": "strong").

value = links x replicas(all ranks) x K / wall-seconds (max over ranks) of the timed region; inputs are resident in
HBM before the region starts.  The region: barrier + synchronize | clock starts | K steps | synchronize | clock stops | barrier;
every rank times its own K steps to completion and the job's time is the MAX over ranks.

OUTPUT.  The LAST line of stdout is ONE compact JSON object (< 4 KB: numbers and short identifiers only -- compact_line(); the driver
keeps an 8 KB tail of stdout and parses its last line); everything measured, with the prose that says how each figure was formed, is
written to bench_full.json beside this script.  Both carry
  launch_plan  how pedn_run stepped (pedn_plan_info): chains of launches, whether the link update of step t is performed by the slot
               waves of step t + 1's node kernel (ONE launch per step), how the nodes were packed into workgroups
  roofline     dominant kernel (node_kernel).  `basis_kind` names the basis of `achieved` / `frac`: "moved" = memory-side bytes per
               launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, run as child processes BEFORE this
               process touches the GPU (N = 1) -- the figure the memory system can be held to; "algorithmic" without such passes
               (N > 1, --no-live-traffic): SURVEY 8(d)'s contract bytes per launch (212 B per link-update when the node kernel
               performs the link update too, 164 B otherwise).  Always beside it: `frac_algorithmic`, `frac_counter` (+ `traffic`),
               `frac_executed` = the contract's bytes on the paths this workload really takes (the four diffusion look-backs only
               where get_outflow runs, widths every replica shares as scalar loads, the gate record only where it changes), tallied
               by the oracle over the replicas cpu_baseline ran; `fits_infinity_cache`: whether "hbm" means HBM for this batch;
               the whole step as `whole_step_frac` (212 B per link-update) / `whole_step_frac_counter`
  cpu_baseline the C restatement under oracle/ timed on this host's cores (1 thread and all cores, CPU model stated) on a
               bounded sample of the same workload + the derived reference-Python equivalent (profiles/cpu_calibration.json)
  extra        measured in the same run (N = 1 only; --no-extra skips them; one figure per config in the compact line): the headline over
               300 steps after 100 when the command's own window is shorter; BASELINE config #3 (delft x 1024); the headline
               network at 4096 replicas (working set beyond the Infinity Cache) with its own counter passes; the headline network
               under 12 x demand; BASELINE config #2's shape (nine_intersections x 256); BASELINE config #5 (45_intersections,
               batched RL step, env-steps/s) with a shared scenario and with per-env randomised ones, at 1024 ... 8192 envs, SUSTAINED
               over whole episodes with the resets inside the timed region, and END TO END with a policy in the loop (random / MLP;
               host-synchronised, chained by events, and replayed as ONE captured graph per four policy steps)

`python bench.py --gpus N` with N > 1 and no RANK in the environment starts its N ranks itself (torch.distributed.run as a
child process, before anything touches the GPU) and exits with the child's code; a WORLD_SIZE that disagrees with --gpus
is an error.
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


def cpu_model_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle_rate(model, net, origin_nodes, threads, seconds_target, key0=0, demand_scale=1.0):
    """Oracle (C restatement) on `threads` host threads, one replica per thread, full episodes, for ~seconds_target.
    Also returns the oracle's tallies of the paths cal_sending_flow took (summed over the replicas run)."""
    import oracle_driver as od

    T = int(model["T"])
    oracles = [od.Oracle(model, seed=0, replica=i) for i in range(threads)]
    done, t0 = 0, time.perf_counter()
    rounds = 0
    while True:
        for i, o in enumerate(oracles):
            key = key0 + rounds * threads + i
            o.reset(seed=0, replica=key)
            for nid in origin_nodes:
                o.set_demand(net.nodes[nid].index, replica_demand(T, key, base=5.0 * demand_scale, peak=10.0 * demand_scale))
        od.run_many(oracles, 1, T)
        done += threads
        rounds += 1
        el = time.perf_counter() - t0
        if el >= seconds_target or rounds >= 64:
            break
    tally = np.sum([o.tally() for o in oracles], axis=0)
    for o in oracles:
        o.close()
    return done * (T - 1) * int(model["n_links"]) / el, done, el, tally


def executed_bytes(tally, shared_widths=True, gate_changes=False):
    """SURVEY 8(d)'s 212 B per link-update restricted to the paths the workload takes.  The contract counts, per link-update: the
    four inflow look-backs of get_outflow (32 B) -- read only where the sending flow is positive on a free-flowing link
    (tally[3] of tally[0] calls); front and back gate widths (16 B) -- one scalar load per wave when every replica shares them;
    the gate record (8 B written) -- stored only where it differs from the link's width; density[t'] (4 B) -- recomputed from
    num_pedestrians with the link update's own division, never read; num_pedestrians of the reverse link twice (sending and
    receiving side, 8 B) -- read once."""
    calls = float(tally[0])
    p_diff = float(tally[3]) / calls if calls else 0.0
    b = 212.0 - 32.0 * (1.0 - p_diff) - 4.0 - 4.0
    if shared_widths:
        b -= 16.0
    if not gate_changes:
        b -= 8.0
    return b, {"sending_flow_calls": int(tally[0]), "past_free_flow_gate": int(tally[1]), "positive_before_draw": int(tally[2]),
               "diffusion_lookbacks_read": int(tally[3]), "activity_draws": int(tally[4]), "p_diffusion": p_diff}


def cpu_baseline(model, net, origin_nodes, network, seconds_target=7.0, demand_scale=1.0):
    """the C restatement on this host, 1 thread and all cores, with the CPU named; and the
    reference-Python equivalent derived through the ratio tools/calibrate_cpu.py measured where the reference lives."""
    cores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = cores
    T = int(model["T"])
    one, n1, s1, t1 = _oracle_rate(model, net, origin_nodes, 1, seconds_target, demand_scale=demand_scale)
    alln, na, sa, ta = _oracle_rate(model, net, origin_nodes, usable, seconds_target, key0=10000, demand_scale=demand_scale)
    ex_bytes, ex_detail = executed_bytes(t1 + ta)
    out = {"value": alln, "unit": "link-updates/s", "cores": usable, "kind": "port",
           "sample_short": f"{na} replicas x {T - 1} steps of {network}, {usable} threads, {sa:.1f} s; 1 thread: {n1} replicas, {s1:.1f} s",
           "sample": f"{na} replicas x {T - 1} steps of {network} on {usable} host threads ({sa:.1f} s) and {n1} replicas on 1 thread "
                     f"({s1:.1f} s); oracle/pedn_oracle.c, one replica per thread, same per-replica demand and RNG keys as the GPU run",
           "one_thread": one, "all_cores": alln, "cores_total": cores, "cores_usable": usable, "cpu_model": cpu_model_name(),
           # which paths of cal_sending_flow those replicas took over their whole episodes (engine == oracle bit for bit, so these are
           # the GPU's own branch statistics for the same replicas): the contract's 212 B on executed paths
           "algorithmic_bytes_executed_per_link_update": ex_bytes, "executed_paths": ex_detail}
    cal = os.path.join(ROOT, "profiles", "cpu_calibration.json")
    if os.path.exists(cal):
        with open(cal) as f:
            c = json.load(f)
        k = c.get("networks", {}).get(network)
        if k:
            ratio = k["port_over_reference_1core"]
            out["reference_python_equiv"] = {
                "label": "derived, not measured here: port rate on this host / (port / reference ratio measured in the build container)",
                "one_core": one / ratio, "all_cores": alln / ratio, "port_over_reference_1core": ratio,
                "calibration": "profiles/cpu_calibration.json (tools/calibrate_cpu.py)", "calibration_host": c.get("host", {})}
    return out


LIVE_TRAFFIC = {}     # (network, replicas) -> {"node_kernel": bytes per launch, ...} measured by live_traffic() before anything touched the GPU


def live_traffic(workloads, history):
    """Memory-side bytes per launch of the step kernels, measured NOW: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes
    (separate runs, as MI355X_MICROARCH.md prescribes) of a short run of this same command as child processes, plus the two
    calibration launches of tools/pmc_calibrate.py that fix the gfx950 unit of FETCH_SIZE.  workloads: (network, replicas)
    pairs.  Must run before this process
    initialises the GPU (a child of a process that holds the GPU must not exec).  Any failure leaves LIVE_TRAFFIC empty and
    the line falls back to the committed profiles/rNN_pmc.json, labelled as such."""
    import csv
    import glob
    import shutil
    import signal
    import statistics
    import subprocess
    import tempfile

    roc = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(roc) or os.environ.get("PEDN_BENCH_CHILD"):
        return
    tmp = tempfile.mkdtemp(prefix="pedn_pmc_", dir="/tmp")
    # PEDN_STREAM_PROBE=0: counter collection serialises kernels, so the probe of the two-chain plan would see no overlap and fall back
    # to one chain -- whole-batch launches, not the half-batch ones of the timed run whose bytes these passes are meant to count
    env = dict(os.environ, PEDN_BENCH_CHILD="1", TMPDIR="/tmp", PEDN_STREAM_PROBE="0")
    deadline = time.perf_counter() + float(os.environ.get("PEDN_BENCH_PMC_BUDGET_S", "240"))   # for ALL passes: the headline line must not wait on a slow profiler

    def counters(counter, tag, cmd):
        d = os.path.join(tmp, tag)
        # own process group: a pass that hangs is killed together with the program under the profiler, so that nothing of it is
        # left on the GPU when the timed run starts
        proc = subprocess.Popen([roc, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + cmd, cwd="/tmp", env=env,
                                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
        left = deadline - time.perf_counter()
        try:
            rc = proc.wait(timeout=max(5.0, min(120.0, left)))
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)
            proc.wait()
            raise
        if rc != 0:
            raise RuntimeError(f"rocprofv3 --pmc {counter} exited with {rc}")
        by = {}
        for r in csv.DictReader(open(glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))[0])):
            name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "")
            by.setdefault(name, []).append(float(r["Counter_Value"]))
        return by

    try:
        cal = [sys.executable, os.path.join(ROOT, "tools", "pmc_calibrate.py")]
        fetch_factor = round(2 * (1 << 26) * 8 / (counters("FETCH_SIZE", "cf", cal)["device_math_kernel"][0] * 1024), 2)
        write_factor = round((1 << 26) * 8 / (counters("WRITE_SIZE", "cw", cal)["device_math_kernel"][0] * 1024), 2)
        for network, replicas in workloads:      # the headline first; the extras only while the budget lasts
            if deadline - time.perf_counter() < 45:
                print(f"bench.py: PMC budget spent, no live passes for {network} x {replicas}", file=sys.stderr)
                break
            cmd = [sys.executable, os.path.abspath(__file__), "--network", network, "--replicas", str(replicas), "--history", history,
                   "--steps", "48", "--warmup", "20", "--no-cpu-baseline", "--no-extra"]
            f, w = counters("FETCH_SIZE", f"pf_{network}_{replicas}", cmd), counters("WRITE_SIZE", f"pw_{network}_{replicas}", cmd)
            out = {"fetch_factor": fetch_factor, "write_factor": write_factor}
            for k in f:
                if k in ("node_kernel", "link_kernel", "link_kernel_1r", "link_turn_kernel") and k in w:
                    skip = len(f[k]) // 4          # the warm-up launches (and the first, cold ones)
                    out[k] = statistics.mean(f[k][skip:]) * 1024 * fetch_factor + statistics.mean(w[k][skip:]) * 1024 * write_factor
            LIVE_TRAFFIC[(network, replicas)] = out
    except Exception as exc:    # noqa: BLE001 -- a profiler that is missing or refuses must not cost the bench line
        print(f"bench.py: live PMC passes failed ({type(exc).__name__}: {exc}); traffic falls back to the committed summary", file=sys.stderr)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measured_traffic(kernel="node_kernel", network="melbourne", replicas=1024, plan=None):
    """Memory-side bytes per launch from the committed rocprofv3 PMC passes (profiles/rNN[_network]_pmc.json, written by
    tools/summarize_profiles.py from separate --pmc FETCH_SIZE / WRITE_SIZE runs of this same command).  bench.py cannot
    collect PMC counters on itself: the number rides along, labelled with its file, only for the workload it was measured on."""
    import glob
    if (network, replicas) in LIVE_TRAFFIC and kernel in LIVE_TRAFFIC[(network, replicas)]:
        lt = LIVE_TRAFFIC[(network, replicas)]
        return lt[kernel], (f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command run as child processes before the timed "
                            f"run (FETCH_SIZE x {lt['fetch_factor']}, WRITE_SIZE x {lt['write_factor']} from two calibration launches)")
    if replicas != 1024:
        return None, None
    suffix = "" if network == "melbourne" else f"_{network}"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]{suffix}_pmc.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        d = json.load(f)
    # a committed summary stands in only when it was taken under the launch plan of this run (round 4 changed the default plan: bytes
    # per launch of node_kernel under the owner-wave plan / two chains are not those of earlier rounds)
    if plan is not None and d.get("launch_plan") != {k: plan[k] for k in ("chains", "link_update_by_next_node_kernel")}:
        return None, None
    for name, k in d.get("kernels", {}).items():
        if name.startswith(kernel):
            return k["hbm_bytes_per_launch"], os.path.relpath(files[-1], ROOT)
    return None, None


def measure_rl(network, B, steps, warmup, history, randomized=False):
    """BASELINE config #5: env-steps/s of the batched RL step (action clipping + network_loading + observations + rewards)
    on `network` x B envs, obs option3, action_gap 1, uniform random actions resident in HBM.  randomized: the envs were reset
    with options={'randomize': True} first (rl/pz_pednet_env.py:143-193, src/utils/env_loader.py:160-181): every env carries its
    own link parameters, OD weights and demand -- the per-replica-parameter kernels."""
    import torch

    from pednstream_amd.rl_env import VecPedNetEnv

    env = VecPedNetEnv(network, n_envs=B, obs_mode="option3", action_gap=1, seed=0, data_dir=os.path.join(ROOT, "data"), history=history)
    e = env.network.engine()
    reset_s = None
    if randomized:
        np.random.seed(0)
        t0 = time.perf_counter()
        env.reset(options={"randomize": True}, seed=0)
        e.synchronize()
        reset_s = time.perf_counter() - t0
    T = env.simulation_steps
    K = min(steps, T - 1 - warmup)
    gen = torch.Generator(device="cuda").manual_seed(0)
    hi = torch.as_tensor(env.action_high, device="cuda", dtype=torch.float64)
    acts = torch.rand((warmup + K, B, env.n_actions), generator=gen, device="cuda", dtype=torch.float64) * hi
    torch.cuda.synchronize()
    row = B * env.n_actions * 8
    t = 1
    for k in range(warmup):
        e.rl_step_device(acts.data_ptr() + k * row, t)
        t += 1
    e.synchronize()
    e.timer_begin()
    t0 = time.perf_counter()
    for k in range(warmup, warmup + K):
        e.rl_step_device(acts.data_ptr() + k * row, t)
        t += 1
    dev_ms = e.timer_end()
    wall = time.perf_counter() - t0
    rc, _ = e.error_flags()
    assert rc == 0
    L = e.n_links
    out = {"metric": "env-steps/sec (replicas x steps/sec, incl. action apply, observations, rewards)", "value": B * K / wall,
           "unit": "env-steps/s", "n_gpus": 1, "steps": K, "warmup": warmup, "ms_per_step": wall / K * 1e3,
           "device_ms_per_step": dev_ms / K, "higher_is_better": True, "data": "synthetic", "dtype": "f64+f32",
           "link_updates_per_s": L * B * K / wall,
           # the whole env step against the roofline: SURVEY 8(d)'s 212 B per link-update (the observation / reward / action rows add
           # 0.3 % on this network and are left out) over the device time of the timed region
           "whole_step_frac": BYTES_PER_LINK_UPDATE * L * B / (dev_ms / K * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "launches_per_env_step": 2,
           "randomized": bool(randomized),
           "config": {"workload": f"{network} x {B} envs, obs option3 ({env.n_obs} floats), {env.n_actions} action dims, "
                                  f"agents {env.possible_agents}, actions resident in HBM (torch), obs/rewards left on device"
                                  + (", every env with its own randomised scenario (reset(options={'randomize': True}))" if randomized else ""),
                      "history": history,
                      "history_bytes": int(sum(e.history_rows(f) * (e.n_all if f < 4 else e.n_links) * ((B + 127) // 128 * 128) * (8 if f < 7 else 4)
                                               for f in range(13)))}}
    if reset_s is not None:
        out["randomized_reset_s"] = reset_s
    env.close()
    return out


def sustained_rl(network, B, history, mode, episodes=3):
    """Config #5 as a training loop runs it: `episodes` whole episodes of T - 1 env steps each, every one started by a reset INSIDE the
    timed region (rl/pz_pednet_env.py:143-193): mode "plain" = reset(); "vectorised" = reset(options={'randomize': True, 'mode':
    'vectorised'}) -- every env a new scenario drawn on the device (pedn_randomize_scenarios); "reference" = the same with the
    reference's randomisers, one np.random stream consumed env after env on the host (bit-exact goldens, host-bound)."""
    import torch

    from pednstream_amd.rl_env import VecPedNetEnv

    env = VecPedNetEnv(network, n_envs=B, obs_mode="option3", action_gap=1, seed=0, data_dir=os.path.join(ROOT, "data"), history=history)
    e = env.network.engine()
    T = env.simulation_steps
    K = T - 1
    gen = torch.Generator(device="cuda").manual_seed(1)
    hi = torch.as_tensor(env.action_high, device="cuda", dtype=torch.float64)
    acts = torch.rand((K, B, env.n_actions), generator=gen, device="cuda", dtype=torch.float64) * hi
    torch.cuda.synchronize()
    row = B * env.n_actions * 8
    options = None if mode == "plain" else {"randomize": True, "mode": "vectorised" if mode == "vectorised" else "reference"}

    def episode(seed):
        e.synchronize()               # the previous episode's steps were enqueued asynchronously: let them finish before timing the reset
        t0 = time.perf_counter()
        env.reset(options=options, seed=seed)
        e.synchronize()
        t_reset = time.perf_counter() - t0
        for k in range(K):
            e.rl_step_device(acts.data_ptr() + k * row, k + 1)
        return t_reset

    np.random.seed(0)
    episode(100)                      # warm-up episode (first use of the per-replica kernels, allocations)
    e.synchronize()
    t0 = time.perf_counter()
    resets = [episode(101 + i) for i in range(episodes)]
    e.synchronize()
    wall = time.perf_counter() - t0
    rc, _ = e.error_flags()
    assert rc == 0
    env.close()
    return {"value": B * K * episodes / wall, "unit": "env-steps/s", "episodes": episodes, "env_steps_per_episode": K, "n_envs": B,
            "wall_s": wall, "reset_ms_mean": float(np.mean(resets)) * 1e3, "reset_share_of_wall": float(np.sum(resets)) / wall,
            "history": history, "reset": mode}


def rl_end_to_end(network, B, steps=600):
    """Config #5 the way BASELINE words it -- the envs FEEDING a policy: policy forward -> env step -> reward bookkeeping per iteration,
    everything on the GPU, env-steps/s END TO END (host enqueue included) over `steps` iterations of one episode.  Two policies: a
    random one (one torch.rand per step) and a 3-layer MLP on the observations (float32, 64 hidden units).  Three ways of running the
    loop: step_device synchronising the host every step; streams chained by events (sync=False); and the whole iteration captured once
    as a torch.cuda.CUDAGraph and replayed (VecPedNetEnv.capture, four policy steps per graph: the engine's device-resident step clock
    makes the env step's launches constant).  What bounds the replayed loop: profiles/r05_graph_rollout.txt."""
    import torch

    from pednstream_amd.rl_env import VecPedNetEnv

    env = VecPedNetEnv(network, n_envs=B, obs_mode="option3", action_gap=1, seed=0, data_dir=os.path.join(ROOT, "data"), history="recent")
    low = torch.as_tensor(env.action_low, device="cuda", dtype=torch.float64)
    span = torch.as_tensor(env.action_high, device="cuda", dtype=torch.float64) - low
    gen = torch.Generator(device="cuda").manual_seed(0)
    torch.manual_seed(0)
    mlp = torch.nn.Sequential(torch.nn.Linear(env.n_obs, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                              torch.nn.Linear(64, env.n_actions), torch.nn.Sigmoid()).to("cuda")
    for p in mlp.parameters():
        p.requires_grad_(False)

    def random_policy(obs):
        return low + span * torch.rand((B, env.n_actions), generator=gen, device="cuda", dtype=torch.float64)

    def mlp_policy(obs):
        return (low + span * mlp(obs).double()).contiguous()

    ret = torch.zeros(B, device="cuda")

    def on_step(obs, rew):
        ret.add_(rew[:, 0])

    K = min(steps, env.simulation_steps - 2)
    out = {}
    for pname, policy in (("end_to_end_random_policy", random_policy), ("end_to_end_mlp_policy", mlp_policy)):
        res = {}
        for label in ("host_synchronised_every_step", "streams_chained_by_events", "graph_replay"):
            env.reset()
            ret.zero_()
            obs = env.device_views()[0]
            roll = env.capture(policy, on_step, generators=[gen], steps_per_replay=4) if label == "graph_replay" else None
            for _ in range(8):                        # warm-up inside the episode (the graph is captured here)
                if roll is not None:
                    if env.sim_step < 9:
                        roll.step()
                else:
                    o, r, _ = env.step_device(policy(obs), sync=label == "host_synchronised_every_step")
                    on_step(o, r)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s0 = env.sim_step
            while env.sim_step - s0 < K - 8:
                if roll is not None:
                    roll.step()
                else:
                    o, r, _ = env.step_device(policy(obs), sync=label == "host_synchronised_every_step")
                    on_step(o, r)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            n_it = env.sim_step - s0
            res[label] = {"value": B * n_it / dt, "unit": "env-steps/s", "steps": n_it, "us_per_iteration": dt / n_it * 1e6,
                          "mean_return": float(ret.mean())}
            if roll is not None:
                res[label]["replays"], res[label]["eager_steps"] = roll.replays, roll.eager_steps
        out[pname] = res
    # ... and over WHOLE episodes with their resets inside the timed region (every reset draws a new scenario per env on the device;
    # the first step of an episode runs eagerly, the rest is replays): what a training loop sees per policy step
    roll = env.capture(mlp_policy, on_step, steps_per_replay=4)
    for timed in (False, True):
        torch.cuda.synchronize()
        t0, n_it = time.perf_counter(), 0
        for ep in range(2 if timed else 1):
            env.reset(options={"randomize": True, "mode": "vectorised"}, seed=50 + ep)
            while not roll.step():
                pass
            n_it += env.sim_step - 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    out["end_to_end_mlp_policy"]["graph_replay_whole_episodes_with_randomised_resets"] = {
        "value": B * n_it / dt, "unit": "env-steps/s", "steps": n_it, "us_per_iteration": dt / n_it * 1e6, "episodes": 2,
        "replays": roll.replays, "eager_steps": roll.eager_steps, "recaptures": roll.recaptures}
    env.close()
    return out


def bench_rl(args):
    if args.rl_end_to_end:
        print(json.dumps(rl_end_to_end(args.network, args.replicas, args.steps)), flush=True)
        return
    print(json.dumps(measure_rl(args.network, args.replicas, args.steps, args.warmup, args.history, args.randomize)), flush=True)


def spawn_ranks(args, argv):
    """`python bench.py --gpus N` (N > 1) outside a launcher: start the N ranks as a CHILD process (torch.distributed.run, one
    rank per GPU, rendezvous on 127.0.0.1) before this process has touched the GPU, and hand its exit code back."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd)


def measure(args, network, dist, rank, local_rank, world, demand_scale=1.0):
    """Warm-up, the timed region, per-kernel durations; returns (fields of the JSON line, net, engine-side context).
    demand_scale: the origin demand of every replica times this (12: the congested regime of the goldens melbourne_heavy_*)."""
    from pednstream_amd import NetworkEnvGenerator
    from pednstream_amd.ensemble import shard

    if args.total_replicas:                                    # strong scaling: the ensemble is fixed, every GPU owns total / N replicas
        if args.total_replicas % world:
            raise SystemExit(f"--total-replicas {args.total_replicas} is not a multiple of --gpus {world}")
        R = args.total_replicas // world
    else:
        R = args.replicas                                      # weak scaling: R replicas on every GPU
    offset, count = shard(R * world, world, rank)              # contiguous block of global replica ids
    assert count == R
    gen = NetworkEnvGenerator(os.path.join(ROOT, "data"))
    net = gen.create_network(network, verbose=False, n_replicas=R, replica_offset=offset, rng_seed=0,
                             rng_mode=args.rng_mode, device=local_rank, history=args.history)
    T = net.simulation_steps
    e = net.engine()
    origins = list(net.origin_nodes)
    for nid in origins:                        # one upload per origin: [R, T] rows keyed by the global replica id
        net.set_demand_matrix(nid, np.stack([replica_demand(T, offset + r, base=5.0 * demand_scale, peak=10.0 * demand_scale) for r in range(R)]))
    e.synchronize()
    L = e.n_links

    def barrier():
        e.synchronize()
        if dist is not None:
            import torch
            dist.barrier()
            if args.backend == "nccl":
                torch.cuda.synchronize()
        e.synchronize()

    # the simulation clock runs 1..T-1; an episode that reaches T is reset and continues (reset is inside the timed region)
    state = {"t": 1}

    def advance(n):
        left = n
        while left > 0:
            if state["t"] >= T:
                e.reset(lazy=True)      # an episode turn-over inside the timed region: what a new episode reads before it writes is restored
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
    e.synchronize()
    if dist is not None and args.backend == "nccl":
        import torch
        torch.cuda.synchronize()
    # this rank's K steps have completed: its clock stops HERE, the closing barrier follows, and the MAX over ranks below is the
    # job's time.  (Rounds 1-4 stopped the clock behind the barrier: an RCCL barrier is a small all-reduce plus a stream
    # synchronisation, 0.15-0.25 ms -- a third of a 20-step region of 31 us steps, charged to the step only at N > 1, where it read
    # as a scaling loss of a path that has no collective.)
    wall = time.perf_counter() - t0
    barrier()
    ranks_seen = 1
    if dist is not None:
        import torch
        dev = "cuda" if args.backend == "nccl" else "cpu"
        w = torch.tensor([wall], device=dev, dtype=torch.float64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
        ones = torch.ones(1, device=dev, dtype=torch.float64)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)           # every rank that really took part adds one
        ranks_seen = int(ones.item())
    rc, _ = e.error_flags()
    if rc != 0:
        raise SystemExit(f"model error flags set: {rc}")

    # per-launch durations (the dispatches' own start / stop timestamps) under the SAME launch plan as the timed region, continuing
    # the same simulation.  chains = 2: run() launched the two halves of the replicas as two chains on two streams; a launch then
    # covers R / 2 replicas and overlaps the other chain's launches.
    def profile(n):
        """Per-launch mean durations (dispatch timestamps) of the three kinds of launch, and the number of chains."""
        if state["t"] + n >= T - 1:
            e.reset()
            e.step(1)                 # the first step of an episode carries the stand-alone turning-fraction launch
            state["t"] = 2
        rows, ch = e.profile_timeline(state["t"], state["t"] + n)
        state["t"] += n
        ms = [float(np.mean(rows[rows[:, 2] == k, 4] - rows[rows[:, 2] == k, 3])) if np.any(rows[:, 2] == k) else 0.0 for k in range(3)]
        return ms, ch

    n_prof = min(40, max(8, args.steps // 8))
    (tf_ms, node_ms, link_ms), chains = profile(n_prof)
    plan = e.plan_info()
    owner = plan["link_update_by_next_node_kernel"]
    if owner:
        # owner-wave plan: node_kernel<LU>(t + 1) performs the link update of t; the range's ONE trailing link_kernel per chain is
        # amortised over its steps.  The dominant kernel then carries the whole step's contract bytes.
        link_ms = link_ms / n_prof
    node_kernel_bytes = BYTES_PER_LINK_UPDATE if owner else NODE_KERNEL_BYTES
    one_chain = None
    if chains == 2 and not args.no_extra:
        # the same kernels launched over the whole batch, one chain on one stream: what a launch achieves on its own
        e.set_streams(1)
        (tf1, node1, link1), _ = profile(n_prof)
        if owner:
            link1 = link1 / n_prof
        e.set_streams(2)
        one_chain = {"avg_launch_ms": float(node1), "frac": node_kernel_bytes * L * R / (node1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "other_kernels_ms": {"link_kernel(+turn_frac of t+1)": float(link1), "turn_frac_kernel(stand-alone)": float(tf1)},
                     "whole_step_frac": BYTES_PER_LINK_UPDATE * L * R / ((node1 + link1 + tf1) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "note": "diagnostic: pedn_set_streams(1), every launch covers all replicas and runs alone"}

    alt_plan = None
    if chains == 1 and not args.no_extra and world == 1 and R % 256 == 0:
        # the other launch plan on the same engine: the two halves of the replicas as two chains of launches on two streams
        e.set_streams(2)
        e.reset()                     # the same steps of a fresh episode as the timed region above
        state["t"] = 1
        advance(args.warmup)
        barrier()
        e.timer_begin()
        t1 = time.perf_counter()
        advance(args.steps)
        dev2 = e.timer_end()
        barrier()
        wall2 = time.perf_counter() - t1
        e.set_streams(1)
        alt_plan = {"plan": "pedn_set_streams(2): two chains of launches, R / 2 replicas each, on two streams", "value": L * R * args.steps / wall2,
                    "unit": "link-updates/s", "ms_per_step": wall2 / args.steps * 1e3, "device_ms_per_step": dev2 / args.steps,
                    "whole_step_frac_wall": BYTES_PER_LINK_UPDATE * L * R / (wall2 / args.steps) / 1e9 / HBM_PEAK_GBS}

    total_lu = L * R * world * args.steps
    node_bytes = node_kernel_bytes * L * R // chains          # one launch covers R / chains replicas
    step_bytes = BYTES_PER_LINK_UPDATE * L * R
    # time the machine spends on one step of all replicas: one chain -> the sum of its launches (gaps excluded); two chains ->
    # their launches overlap, so the device time of the timed region (HIP events around it) per step
    step_ms = node_ms + link_ms + tf_ms if chains == 1 else dev_ms / args.steps
    # The dominant kernel against the roofline: its contract bytes of one step (all chains) over ITS SHARE of the step's machine time =
    # step_ms x (its launch duration / the sum of the step's launch durations).  One chain: step_ms is that sum, so this is bytes per
    # launch / mean launch duration.  Two chains: the launches of the two chains overlap and share the machine, so a launch's own
    # duration describes half a machine; the unperturbed device time per step is apportioned among the kernels by their durations.
    node_busy_ms = step_ms * node_ms / (node_ms + link_ms + tf_ms)
    achieved_alg = node_kernel_bytes * L * R / (node_busy_ms * 1e-3) / 1e9
    traffic, traffic_src = measured_traffic("node_kernel", network, R, plan)     # per launch, measured under this same plan
    live = traffic is not None and traffic_src.startswith("live")
    # the second launch's memory-side bytes, for the working set of a step
    traffic2 = None
    for k2 in ("link_turn_kernel", "link_kernel_1r", "link_kernel"):
        t2, _ = measured_traffic(k2, network, R, plan)
        if t2 is not None:
            traffic2 = t2
            break
    # `achieved` / `frac` stay on ONE basis in every run and on every rank: the contract's algorithmic bytes per launch over the launch's
    # duration (`basis_kind`).  What the memory side MOVED for the launch (FETCH_SIZE + WRITE_SIZE) rides beside it as `traffic` /
    # `frac_counter` (null without counter passes, live or committed), and the contract's bytes on executed paths as `frac_executed`
    # (filled in from cpu_baseline's tallies).
    # `frac` / `achieved` are on the bytes the memory side MOVED whenever this run's own counter passes exist (`basis_kind`: "moved" --
    # the figure the memory system can be held to); without counters they are the contract's bytes ("algorithmic").  Both always ride
    # along as frac_counter / frac_algorithmic, and the contract's bytes on executed paths as frac_executed.
    achieved = achieved_alg if not live else traffic * chains / (node_busy_ms * 1e-3) / 1e9
    if owner:
        traffic2 = 0.0               # one launch per step: the range's single trailing link_kernel is not part of a step's working set
    working_set = None if traffic is None or traffic2 is None else (traffic + traffic2) * chains
    mdl = e.model
    n_dyn_turns = int(np.diff(mdl["node_turn_ptr"])[np.asarray(mdl["node_dyn"]) > 0].sum())     # +8 B each per replica
    out = {
        "metric": "link-updates/sec (links x replicas x steps/sec)", "value": total_lu / wall, "unit": "link-updates/s",
        "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if args.total_replicas else "weak", "vs_baseline": None, "dtype": "f64+f32", "data": "synthetic",
        "config": {"workload": f"{network} network ({L} links, {len(net.nodes)} nodes, T={T}) x {R} replicas per GPU, "
                               f"{'full-record' if args.history == 'full' else 'recent-history'} mode, per-replica Poisson demand and Philox keys",
                   "network": network, "history": "full-record" if args.history == "full" else "recent",
                   "replicas_per_gpu": R, "replicas_total": R * world, "links": L,
                   "parallelism": f"replica-sharded x{world}, no step-path collective"},
        "device_ms_per_step": dev_ms / args.steps,
        "roofline": {"bound": "hbm",
                     "kernel": ("node_kernel<LU>: sending / receiving flows, node model, cumulative counts AND the link update of the previous step "
                                "(one launch per step)" if owner else "node_kernel: sending / receiving flows, node model, cumulative counts"),
                     "kernel_name": "node_kernel<LU>" if owner else "node_kernel",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "basis_kind": "moved" if live else "algorithmic",
                     "basis": f"SURVEY 8(d) contract bytes of the kernel's launches of one step ({node_kernel_bytes} B per link-update x {L * R} "
                              "link-updates) / the kernel's share of the step's machine time = device time per step x (its mean launch duration / "
                              "the sum of the step's mean launch durations; dispatch timestamps, pedn_profile_timeline).  With one chain of launches "
                              "that is bytes per launch / mean launch duration; with two chains the launches overlap and share the machine.  The "
                              "same basis in every run; the bytes the memory side moved are `traffic` / `frac_counter`, the contract's bytes on "
                              "executed paths `frac_executed`",
                     "kernel_busy_ms_per_step": float(node_busy_ms),
                     "per_launch": {"algorithmic_bytes": node_bytes, "avg_ms": float(node_ms), "GBps": node_bytes / (node_ms * 1e-3) / 1e9,
                                    "note": "one launch = replicas / concurrent_chains; under two chains it shares the machine with the other chain's launch"},
                     "achieved_algorithmic": achieved_alg, "frac_algorithmic": achieved_alg / HBM_PEAK_GBS,
                     "traffic_is_live": bool(live),
                     "traffic": traffic, "traffic_source": traffic_src,
                     "frac_counter": None if traffic is None else traffic * chains / (node_busy_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "traffic_bytes_per_link_update": None if traffic is None else traffic * chains / (L * R),
                     # what one step touches, and whether "hbm" means HBM: a step whose working set fits the 256 MB Infinity Cache is
                     # served from it between the launches (FETCH_SIZE counts those hits too)
                     "working_set_bytes_per_step": working_set,
                     "fits_infinity_cache": None if working_set is None else bool(working_set < 256e6),
                     "concurrent_chains": chains, "replicas_per_launch": R // chains,
                     "note": ("run() launches the two halves of the replicas as two chains on two streams: `achieved` / `frac` are per launch "
                              "(R / 2 replicas) while the other chain's launches share the machine; the machine-level figure is whole_step_frac"
                              if chains == 2 else "one chain of launches"),
                     "algorithmic_bytes_per_launch": node_bytes, "algorithmic_bytes_per_link_update": node_kernel_bytes,
                     "launch_plan": plan,
                     "avg_launch_ms": float(node_ms),
                     "other_kernels_ms": {"link_kernel(+turn_frac of t+1)": float(link_ms), "turn_frac_kernel(stand-alone)": float(tf_ms)},
                     "second_launch_traffic": traffic2,
                     "whole_step_GBps": step_bytes / (step_ms * 1e-3) / 1e9,
                     "whole_step_frac": step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "whole_step_frac_counter": None if working_set is None else working_set / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "whole_step_basis": "sum of the step's launch durations" if chains == 1 else "device time of the timed region per step (two overlapping chains)",
                     "whole_step_frac_wall": step_bytes / (wall / args.steps) / 1e9 / HBM_PEAK_GBS,
                     "whole_step_bytes_per_link_update": BYTES_PER_LINK_UPDATE,
                     "dynamic_turns": n_dyn_turns,
                     "whole_step_frac_incl_dynamic_turns": (step_bytes + 8 * n_dyn_turns * R) / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
    }
    if one_chain is not None:
        out["roofline"]["one_chain"] = one_chain
    if alt_plan is not None:
        out["two_chain_plan"] = alt_plan
    return out, net, origins


def _sig(x, n=6):
    """Floats to n significant digits (the compact line); everything else unchanged."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float(f"{x:.{n}g}")


def _pick(d, keys):
    return {k: _sig(d[k]) for k in keys if isinstance(d, dict) and k in d and not isinstance(d[k], (dict, list))}


def compact_line(out):
    """The line the driver parses: the LAST line of stdout, < 4 KB, numbers and short identifiers only.  Everything measured (every
    extra with its own roofline block, the prose that says how each figure was formed) is written to bench_full.json beside this script."""
    rf, cb = out["roofline"], out.get("cpu_baseline")
    cfg = out["config"]
    line = {k: _sig(out[k]) for k in ("metric", "value", "unit", "n_gpus", "ranks_seen", "steps", "warmup", "ms_per_step", "device_ms_per_step",
                                      "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in out}
    line["config"] = {"workload": f"{cfg['network']} x {cfg['replicas_per_gpu']} replicas per GPU, {cfg['history']} histories",
                      "replicas_per_gpu": cfg["replicas_per_gpu"], "replicas_total": cfg["replicas_total"], "links": cfg["links"],
                      "parallelism": f"replicas x{out['n_gpus']}"}
    r = _pick(rf, ("bound", "kernel_name", "achieved", "peak", "unit", "frac", "basis_kind", "traffic", "frac_algorithmic", "frac_counter",
                   "frac_executed", "traffic_bytes_per_link_update", "algorithmic_bytes_per_link_update", "algorithmic_bytes_per_launch",
                   "fits_infinity_cache", "concurrent_chains", "avg_launch_ms", "whole_step_frac", "whole_step_frac_counter"))
    r["kernel"] = r.pop("kernel_name")
    pl = rf.get("launch_plan") or {}
    r["launch_plan"] = {k: pl[k] for k in ("chains", "link_update_by_next_node_kernel", "single_launch_step") if k in pl}
    if "one_chain" in rf:
        r["one_chain"] = _pick(rf["one_chain"], ("avg_launch_ms", "frac"))
    line["roofline"] = r
    if cb is not None:
        c = _pick(cb, ("value", "unit", "cores", "kind", "sample_short", "cpu_model", "one_thread", "all_cores"))
        c["sample"] = c.pop("sample_short")
        if "reference_python_equiv" in cb:
            c["reference_python_equiv"] = {"one_core": _sig(cb["reference_python_equiv"]["one_core"])}
        line["cpu_baseline"] = c
    ex = out.get("extra")
    if ex:
        e2 = {}
        for name, v in ex.items():
            if "error" in v:
                e2[name] = {"error": str(v["error"])[:80]}
            elif name == "config5_rl_45int_x2048":
                c5 = {"unit": "env-steps/s"}
                for k in ("plain", "randomized"):
                    if k in v:
                        c5[k] = _pick(v[k], ("value", "ms_per_step", "whole_step_frac")) if "error" not in v[k] else {"error": str(v[k]["error"])[:60]}
                if "by_n_envs_recent_history" in v and "error" not in v["by_n_envs_recent_history"]:
                    c5["by_n_envs"] = {n: _sig(x["value"], 4) for n, x in v["by_n_envs_recent_history"].items()}
                for k in ("end_to_end_random_policy", "end_to_end_mlp_policy"):
                    if k in v:
                        c5[k] = {kk: _sig(x["value"], 4) for kk, x in v[k].items() if isinstance(x, dict) and "value" in x}
                if "sustained" in v and "error" not in v["sustained"]:
                    c5["sustained"] = {kk: _sig(x["value"], 4) for kk, x in v["sustained"].items()}
                e2[name] = c5
            else:
                e2[name] = _pick(v, ("value", "ms_per_step"))
                if "roofline" in v:
                    e2[name].update(_pick(v["roofline"], ("whole_step_frac", "whole_step_frac_counter")))
        line["extra"] = e2
    line["full"] = "bench_full.json"
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--network", default="melbourne")
    ap.add_argument("--replicas", type=int, default=1024, help="replicas per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not run the rocprofv3 --pmc child passes; roofline.traffic then comes from profiles/")
    ap.add_argument("--no-extra", action="store_true", help="skip the delft x 1024 (BASELINE config #3) measurement that rides along at N = 1")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                    "multi-rank path on one GPU)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--rng-mode", default="philox", choices=["philox", "meanfield"], help="diagnostic: meanfield removes the RNG work")
    ap.add_argument("--rl", action="store_true", help="config #5 instead: batched RL env step, env-steps/s")
    ap.add_argument("--randomize", action="store_true", help="with --rl: reset(options={'randomize': True}) first (per-replica scenarios)")
    ap.add_argument("--rl-end-to-end", action="store_true", help="with --rl: the policy-in-the-loop measurement (eager / events / replayed graph) instead")
    ap.add_argument("--total-replicas", type=int, default=0, help="strong scaling: this many replicas in all, total / N per GPU "
                    "(BASELINE config #4: --total-replicas 4096; its weak-scaling shape is --replicas 512)")
    ap.add_argument("--history", default="full", choices=["full", "recent"], help="full: the reference's footprint (the headline mode); "
                    "recent: rings for everything the recurrence does not look far back into (include/pedn.h PEDN_HIST_RECENT)")
    args = ap.parse_args()
    if args.rl:
        return bench_rl(args)

    if "RANK" not in os.environ:
        if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}")
        if args.gpus > 1:             # nothing has touched the GPU yet: the ranks run in a child, this process only relays its exit code
            raise SystemExit(spawn_ranks(args, sys.argv[1:]))
        if args.gpus == 1 and not args.no_extra and not args.rl and not args.no_live_traffic:
            headline = args.network == "melbourne" and args.replicas == 1024 and not args.total_replicas
            live_traffic([(args.network, args.total_replicas or args.replicas)] + ([("delft", 1024), ("melbourne", 4096)] if headline else []), args.history)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.share_device:
        local_rank = 0
        args.backend = "gloo"         # RCCL refuses two ranks on one device; the rehearsal only needs the barrier / reductions
    dist = None
    if "RANK" in os.environ:          # launched by torch.distributed.run, also for N = 1
        import torch
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)

    from pednstream_amd.flatten import flatten_network

    out, net, origins = measure(args, args.network, dist, rank, local_rank, world)
    if out["ranks_seen"] != world:
        raise SystemExit(f"only {out['ranks_seen']} of {world} ranks took part")

    def add_executed(line, cb):
        """The contract's bytes on the paths the workload takes (cpu_baseline's tallies) next to the contract figure."""
        bx = cb["algorithmic_bytes_executed_per_link_update"]
        rf = line["roofline"]
        rf["algorithmic_bytes_executed_per_link_update"] = bx
        rf["frac_executed"] = rf["whole_step_frac"] * bx / BYTES_PER_LINK_UPDATE
        rf["executed_basis"] = ("whole step: 212 B per link-update minus the bytes of paths this workload does not take (bench.py: executed_bytes; "
                                "tallied by the oracle over the replicas cpu_baseline ran, whole episodes) over the step's device time")

    if rank == 0 and world == 1:
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(flatten_network(net), net, origins, args.network)
            add_executed(out, out["cpu_baseline"])
    net.close()
    headline = args.network == "melbourne" and args.replicas == 1024 and not args.total_replicas
    if rank == 0 and world == 1 and not args.no_extra and headline:
        import copy
        keep = ("value", "unit", "ms_per_step", "device_ms_per_step", "steps", "warmup", "config", "roofline")
        out["extra"] = {}

        def extra(name, fn):
            """An extra must never cost the headline line: a failure is reported in its place."""
            try:
                out["extra"][name] = fn()
            except (Exception, SystemExit) as exc:    # noqa: BLE001  (measure() leaves through SystemExit when error flags are set)
                out["extra"][name] = {"error": f"{type(exc).__name__}: {exc}"}

        def variant(network, replicas=1024, steps=None, warmup=None, demand_scale=1.0, cpu_seconds=0.0):
            a2 = copy.copy(args)
            a2.replicas, a2.no_extra = replicas, True
            a2.steps, a2.warmup = steps or args.steps, warmup or args.warmup
            ex, net2, origins2 = measure(a2, network, None, 0, local_rank, 1, demand_scale=demand_scale)
            try:
                if cpu_seconds and not args.no_cpu_baseline:
                    ex["cpu_baseline"] = cpu_baseline(flatten_network(net2), net2, origins2, network, seconds_target=cpu_seconds, demand_scale=demand_scale)
                    add_executed(ex, ex["cpu_baseline"])
            finally:
                net2.close()
            return {k: ex[k] for k in keep + (("cpu_baseline",) if "cpu_baseline" in ex else ())}

        if args.steps < 300:
            # the driver's window (--steps 20 after 5) is a few hundred microseconds at the empty start of an episode: the same
            # measurement over the builder's window, in the same run
            extra("headline_long_window", lambda: variant("melbourne", steps=300, warmup=100))
        # BASELINE config #3 (the network BASELINE.json names for the rocprof roofline), same engine, same run
        extra("config3_delft_x1024", lambda: variant("delft", cpu_seconds=4.0))
        # the same headline network with a working set beyond the 256 MB Infinity Cache (4096 replicas: the per-GPU shape of nothing
        # in BASELINE, but the point where "hbm" means HBM), with its own counter passes
        extra("hbm_proper_melbourne_x4096", lambda: variant("melbourne", replicas=4096, steps=min(args.steps, 120), warmup=min(args.warmup, 30)))
        # the congested regime: 12 x the origin demand per replica (the goldens melbourne_heavy_*): release binomials, get_outflow's
        # look-backs, the congested branch of the sending flow and divergence between lanes all fire
        extra("melbourne_heavy_x1024", lambda: variant("melbourne", steps=max(args.steps, 200), warmup=max(args.warmup, 150), demand_scale=12.0, cpu_seconds=3.0))
        # BASELINE config #2's shape
        extra("nine_x256", lambda: variant("nine_intersections", replicas=256, steps=max(args.steps, 200), warmup=max(args.warmup, 100)))
        # BASELINE config #5: the batched RL env step on 45_intersections, shared scenario and per-env randomised scenarios
        def config5():
            """Every part on its own: a part that fails is reported in its place and costs neither the others nor the headline."""
            c5 = {}

            def part(name, fn):
                try:
                    r = fn()
                    c5.update(r) if name is None else c5.__setitem__(name, r)
                except Exception as exc:    # noqa: BLE001
                    c5[name or "end_to_end"] = {"error": f"{type(exc).__name__}: {exc}"}

            part("plain", lambda: measure_rl("45_intersections", 2048, args.steps, args.warmup, "full", randomized=False))
            part("randomized", lambda: measure_rl("45_intersections", 2048, args.steps, args.warmup, "full", randomized=True))
            # where the env step stops being latency-bound: more envs per launch (recent-history mode: 8192 envs are 16 GB)
            part("by_n_envs_recent_history", lambda: {
                str(n): {k: v for k, v in measure_rl("45_intersections", n, args.steps, args.warmup, "recent", randomized=False).items()
                         if k in ("value", "unit", "device_ms_per_step", "whole_step_frac", "steps")} for n in (1024, 2048, 4096, 8192)})
            # a policy in the loop (torch on the same GPU): what a rollout sees end to end, eager and as a replayed graph
            part(None, lambda: rl_end_to_end("45_intersections", 2048))
            # whole episodes WITH their resets (rl/pz_pednet_env.py:143-193 resets every episode)
            part("sustained", lambda: {f"{mode}_{hist}": sustained_rl("45_intersections", 2048, hist, mode)
                                       for hist in ("full", "recent") for mode in ("plain", "vectorised", "reference")})
            return c5

        extra("config5_rl_45int_x2048", config5)
    if rank == 0:
        # everything measured goes to a file beside the script; the LAST line of stdout is the compact headline object (< 4 KB)
        try:
            with open(os.path.join(ROOT, "bench_full.json"), "w") as f:
                json.dump(out, f)
        except OSError as exc:
            print(f"bench.py: bench_full.json not written ({exc})", file=sys.stderr)
        line = json.dumps(compact_line(out), separators=(",", ":"))
        if len(line) >= 4096:                       # never let detail cost the line: drop the extras before anything else
            slim = compact_line(out)
            slim.pop("extra", None)
            line = json.dumps(slim, separators=(",", ":"))
        print(line, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
