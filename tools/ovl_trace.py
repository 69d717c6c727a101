import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if "miso::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# print a window in the middle
mid = len(rows) // 2
t0 = int(rows[mid]["Start_Timestamp"])
for r in rows[mid:mid + 14]:
    nm = r["Kernel_Name"].split("miso::")[1][:24]
    print(f"{nm:<26} q={r.get('Queue_Id','?'):>3} start={(int(r['Start_Timestamp'])-t0)/1e3:8.1f} end={(int(r['End_Timestamp'])-t0)/1e3:8.1f}")
