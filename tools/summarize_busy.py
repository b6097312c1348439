"""Unit utilisation of the step kernels from two rocprofv3 --pmc passes (SQ_ACTIVE_INST_VALU / _SCA / _ANY, SQ_INSTS_*):
busy % = sum(SQ_ACTIVE_INST_x) / CU_NUM / (mean launch duration * 2.4 GHz), the VALUBusy / SALUBusy expression of
`rocprofv3 -L` with the launch's own duration in place of GRBM_GUI_ACTIVE.

    python tools/summarize_busy.py gpurun_out/sq3 gpurun_out/sq4 > profiles/r02_unit_busy.json
"""
import collections
import csv
import glob
import json
import sys

out = {}
for d in sys.argv[1:]:
    f = glob.glob(f"{d}/*/*_counter_collection.csv")[0]
    kt = glob.glob(f"{d}/*/*_kernel_trace.csv")[0]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(kt)):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, counters in acc.items():
        if "node_kernel" not in k and "link_" not in k and "turn_frac" not in k:
            continue
        name = k.split("(")[0].replace("void ", "")
        e = out.setdefault(name, {"launches": len(dur[k]), "mean_duration_us_under_pmc": round(sum(dur[k]) / len(dur[k]) / 1e3, 2)})
        cyc = sum(dur[k]) / len(dur[k]) * 2.4
        for c, vals in counters.items():
            m = sum(vals) / len(vals)
            e[c] = round(m, 1)
            if c in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS"):
                e[c.replace("SQ_ACTIVE_INST_", "busy_pct_")] = round(100 * m / 256 / cyc, 1)
        if "SQ_INSTS_VALU" in e and "SQ_WAVES" in e:
            e["valu_per_wave"] = round(e["SQ_INSTS_VALU"] / e["SQ_WAVES"], 1)
        if "SQ_INSTS_SALU" in e and "SQ_WAVES" in e:
            e["salu_per_wave"] = round(e["SQ_INSTS_SALU"] / e["SQ_WAVES"], 1)
print(json.dumps(out, indent=1))
