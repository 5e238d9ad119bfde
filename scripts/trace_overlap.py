import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
sel = rows[-14:]
for r in sel:
    print("%-40s q%-3s start %9.1f us  dur %8.1f us" % (r["Kernel_Name"].split("(")[0].replace("void mrs_tg::", "")[:40], r["Queue_Id"], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
