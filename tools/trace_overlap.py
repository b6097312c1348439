"""Overlap of the launches of the two chains, from a rocprofv3 --kernel-trace of a two-chain run (least perturbed view):

    cd /tmp && export TMPDIR=/tmp PEDN_STREAMS=2
    rocprofv3 --kernel-trace --output-format csv -d /tmp/kt2 -- python3 $REPO/bench.py --steps 120 --warmup 40 --no-cpu-baseline --no-extra --no-live-traffic
    python3 $REPO/tools/trace_overlap.py /tmp/kt2

Takes the last 200 step-kernel dispatches, prints a stretch of them (start / end in us, queue) and how much of the time 0 / 1 / 2
of them were in flight."""
import csv
import glob
import sys

import numpy as np

rows = []
for r in csv.DictReader(open(glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0])):
    name = r["Kernel_Name"]
    kind = "node" if "node_kernel" in name else ("second" if ("link_kernel" in name or "link_turn_kernel" in name) else None)
    if kind:
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind, r.get("Queue_Id", "?")))
rows.sort()
rows = rows[-240:-40]
t0 = rows[0][0]
print(f"{len(rows)} dispatches over {(rows[-1][1] - t0) / 1e3:.1f} us; queues: {sorted(set(r[3] for r in rows))}")
print("   kernel  queue      start       end  duration (us)")
for s, e, k, q in rows[100:116]:
    print(f"   {k:7s} {q:>5s} {(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}")
grid = np.arange(0, rows[-1][1] - t0, 100)        # 0.1 us
running = np.zeros(len(grid), dtype=int)
for s, e, k, q in rows:
    running[(grid >= s - t0) & (grid < e - t0)] += 1
for n in range(running.max() + 1):
    print(f"   {n} launches in flight: {100 * (running == n).mean():5.1f} % of the time")
for k in ("node", "second"):
    d = [(e - s) / 1e3 for s, e, kk, q in rows if kk == k]
    print(f"   {k:7s}: mean {np.mean(d):6.2f} us, min {np.min(d):6.2f}, max {np.max(d):6.2f} over {len(d)} dispatches")
steps = len([1 for r in rows if r[2] == "node"]) / 2
print(f"   {(rows[-1][1] - t0) / 1e3 / steps:.2f} us per step of the whole batch")
