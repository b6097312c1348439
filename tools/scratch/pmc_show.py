import csv, glob, sys, collections
for d in sys.argv[1:]:
    f = glob.glob(d + "/*/*_counter_collection.csv")
    if not f: print("no csv in", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        if not any(s in k for s in ("node_kernel", "link_kernel", "turn_prob")): continue
        print(k, {c: round(sum(v[20:]) / max(1, len(v[20:]))) for c, v in cs.items()})
