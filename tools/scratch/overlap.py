import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "node_kernel" in r["Kernel_Name"] or "link_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows)//2: len(rows)//2 + 24]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    print(r["Kernel_Name"][:24].ljust(24), "queue", r.get("Queue_Id"), "start %8.2f  end %8.2f us" % ((int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-t0)/1e3))
