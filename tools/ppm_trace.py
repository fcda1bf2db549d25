import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "ppm_" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 40 dispatches = 8K random or black
big = [r for r in rows if int(r.get("Grid_Size", r.get("Grid_Size_X", "0"))) > 3000000]
print(len(rows), len(big))
prev_end = None
out = []
for r in big[-60:-20]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.append((r["Kernel_Name"][:16], (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0))
    prev_end = e
for o in out[:16]: print(o)
