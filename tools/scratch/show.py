import json, sys
for line in open(sys.argv[1]):
    if line.startswith("=="): print(line.strip())
    elif line.startswith("{"):
        d=json.loads(line); r=d["roofline"]
        print(round(d["ms_per_step"]*1e3,1), "node", round(r["avg_launch_ms"]*1e3,1), {k: round(v*1e3,1) for k,v in r["other_kernels_ms"].items()})
    else: print(line.strip()[:200])
