"""Per-kernel means of the SQ counters of one or more `rocprofv3 --pmc ...` runs (counter_collection.csv) -> JSON on stdout.

  python tools/summarize_sq.py <dir> [<dir> ...] [--skip N]     (N leading launches per kernel are dropped: warm-up)
"""
import csv
import glob
import json
import os
import statistics as st
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
skip = int(sys.argv[sys.argv.index("--skip") + 1]) if "--skip" in sys.argv else 0
args = [a for a in args if a != str(skip) or "--skip" not in sys.argv]
out = {}
for d in args:
    for path in glob.glob(os.path.join(d, "*", "*_counter_collection.csv")):
        per = {}
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"].split("(")[0]
            if name.startswith("void "):
                name = name[5:]
            per.setdefault((name, r["Counter_Name"]), []).append(float(r["Counter_Value"]))
        for (name, ctr), vals in per.items():
            vals = vals[skip:] or vals
            out.setdefault(name, {})[ctr] = st.mean(vals)
            out[name]["launches_averaged"] = len(vals)
for name, c in out.items():
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        c["wait_any_over_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if "SQ_WAVES" in c and c["SQ_WAVES"]:
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_LDS"):
            if k in c:
                c[k + "_per_wave"] = c[k] / c["SQ_WAVES"]
print(json.dumps(out, indent=1))
