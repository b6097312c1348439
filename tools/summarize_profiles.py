"""Condense rocprofv3 outputs merged under gpurun_out/ into the small tracked files under profiles/.

  python tools/summarize_profiles.py <round-tag> <kernel-trace-dir> <pmc-fetch-dir> <pmc-write-dir> <cal-fetch-dir> <cal-write-dir> [warmup] [bench-log]

bench-log: stdout of the profiled bench.py run; its `roofline.launch_plan` goes into the summary, so that bench.py lets the summary
stand in for live counter passes only under the same launch plan (chains, owner-wave link update).

Writes profiles/<tag>_kernel_stats.csv (verbatim rocprofv3 --stats table) and profiles/<tag>_pmc.json with per-kernel
HBM-side bytes per launch.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts half of the bytes of a
coalesced stream (MI355X_MICROARCH.md, HBM section) -- the factor is re-measured here with a known-byte-count launch
(tools/pmc_calibrate.py: 1 GiB read, 0.5 GiB written with 8-byte lanes) instead of being assumed."""
import csv
import glob
import json
import os
import shutil
import statistics as st
import sys

tag, kt, pf, pw, cf, cw = sys.argv[1:7]
warmup = int(sys.argv[7]) if len(sys.argv) > 7 else 100
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "profiles")
os.makedirs(out, exist_ok=True)


def one(pattern):
    return glob.glob(pattern)[0]


shutil.copy(one(os.path.join(kt, "*", "*_kernel_stats.csv")), os.path.join(out, f"{tag}_kernel_stats.csv"))


def counters(d):
    rows = list(csv.DictReader(open(one(os.path.join(d, "*", "*_counter_collection.csv")))))
    by = {}
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].split("<")[0]       # "void node_kernel<false>(DevView, int)" -> "node_kernel"
        if name.startswith("void "):
            name = name[5:]
        by.setdefault(name, []).append(float(r["Counter_Value"]))
    return by


cal_read_bytes, cal_write_bytes = 2 * (1 << 26) * 8, (1 << 26) * 8
cal_f = counters(cf)["device_math_kernel"][0] * 1024
cal_w = counters(cw)["device_math_kernel"][0] * 1024
fetch_factor, write_factor = cal_read_bytes / cal_f, cal_write_bytes / cal_w
f, w = counters(pf), counters(pw)
summary = {"calibration": {"known_read_bytes": cal_read_bytes, "FETCH_SIZE_bytes": cal_f, "fetch_factor": fetch_factor,
                           "known_write_bytes": cal_write_bytes, "WRITE_SIZE_bytes": cal_w, "write_factor": write_factor},
           "kernels": {}}
for k in ("node_kernel", "link_kernel", "link_kernel_1r", "link_turn_kernel", "turn_frac_kernel"):
    if k not in f:
        continue
    fr, wr = f[k][warmup:], w[k][warmup:]
    if not fr or not wr:
        continue
    rd = st.mean(fr) * 1024 * round(fetch_factor, 2)
    wt = st.mean(wr) * 1024 * round(write_factor, 2)
    summary["kernels"][k] = {"launches_averaged": len(fr), "FETCH_SIZE_KiB_mean": st.mean(fr), "WRITE_SIZE_KiB_mean": st.mean(wr),
                             "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wt,
                             "hbm_bytes_per_launch": rd + wt}
if len(sys.argv) > 8 and os.path.exists(sys.argv[8]):
    for line in open(sys.argv[8]):
        if line.startswith("{"):
            plan = json.loads(line)["roofline"].get("launch_plan", {})
            summary["launch_plan"] = {k: plan.get(k) for k in ("chains", "link_update_by_next_node_kernel")}
with open(os.path.join(out, f"{tag}_pmc.json"), "w") as fh:
    json.dump(summary, fh, indent=1)
print(json.dumps(summary, indent=1))
