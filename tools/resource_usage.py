"""Per-kernel register / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage (no GPU needed).

    python tools/resource_usage.py [pattern]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "pednstream_amd", "csrc"), "resource-usage"], capture_output=True, text=True)
rows, cur = [], None
for line in (out.stdout + out.stderr).splitlines():
    m = re.search(r"remark:\s+(.*?)\s*(\[-Rpass|$)", line)
    if not m:
        continue
    txt = m.group(1)
    if txt.startswith("Function Name:"):
        name = subprocess.run(["c++filt", txt.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", name).replace("void ", "")}
        rows.append(cur)
    elif cur is not None and ":" in txt:
        k, v = txt.split(":", 1)
        cur[k.strip()] = v.strip()
pat = sys.argv[1] if len(sys.argv) > 1 else ""
print(f"{'kernel':44s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'spillV':>6s} {'spillS':>6s} {'LDS':>6s} {'occ':>4s}")
for r in rows:
    if pat in r["name"]:
        print(f"{r['name'][:44]:44s} {r.get('VGPRs', '?'):>5s} {r.get('AGPRs', '?'):>5s} {r.get('TotalSGPRs', '?'):>5s} "
              f"{r.get('VGPRs Spill', '?'):>6s} {r.get('SGPRs Spill', '?'):>6s} {r.get('LDS Size [bytes/block]', '?'):>6s} "
              f"{r.get('Occupancy [waves/SIMD]', '?'):>4s}")
